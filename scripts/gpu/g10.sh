set -u
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
OUT=$PWD/gpurun_out/prof_r06; mkdir -p $OUT
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
rocprofv3 --pmc $SQ1 -d "$OUT/pmc_cfg2_sq" -o pmc --output-format csv -- python scripts/pmc_cfg2.py > "$OUT/pmc_cfg2_sq.log" 2>&1
mkdir -p /tmp/s2 && python scripts/profile_summarise.py "$OUT" > "$OUT/summarise2.log" 2>&1
python - <<'PY'
import json
new = json.load(open("gpurun_out/prof_r06/valu.json")); old = json.load(open("profiles/r06/valu.json"))
keys = {(e["kernel"], e["config"]) for e in new}
json.dump([e for e in old if (e["kernel"], e["config"]) not in keys] + new, open("profiles/r06/valu.json", "w"), indent=1)
print("valu.json entries:", [(e["kernel"], e["config"], round(e["valu_per_pass"], 1), round(e["valu_busy"], 3)) for e in json.load(open("profiles/r06/valu.json"))])
PY
cp profiles/r06/valu.json "$OUT/valu_merged.json"
python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "headline: $(cut -c1-200 $OUT/bench.json)"
for c in cfg2 cfg3 cfg4; do python bench.py --config $c --steps 3 --warmup 1 --no-pipelined > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.log"; echo "$c: $(cut -c1-160 $OUT/bench_$c.json)"; done
python bench.py --config cfg3 --batch 65536 --steps 2 --warmup 1 --no-extras > "$OUT/bench_cfg3_64k.json" 2> "$OUT/bench_cfg3_64k.log"
python bench.py --steps 5 --warmup 1 --budget 1500 --no-cpu-baseline > "$OUT/bench_budget1500.json" 2> "$OUT/bench_budget1500.log"
NMPC_BENCH_FORCE_DIST=1 python bench.py --steps 5 --warmup 1 --no-extras > "$OUT/bench_rccl_world1.json" 2> "$OUT/bench_rccl_world1.log"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log"
