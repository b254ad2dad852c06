#!/bin/bash
# Collects the evidence behind bench.py's numbers on the GPU box (one gpurun call; everything lands under gpurun_out/prof_<tag>/):
#   1. PMC passes, each in its own run (no trace domains): FETCH_SIZE | WRITE_SIZE | SQ instruction mix | SQ waits, for the headline batch,
#      for ONE hard instance solved alone (with and without helpers) and for the cfg 2 batch          -> pmc_*.csv, traffic.json
#   1b. executed instructions per evaluation pass by class: basic-block counters in the compiler's own assembly (scripts/bbcount.py)   -> dynamic_mix.json
#   2. how often the exact certificates of eval_psi fall back (a -DNMPC_WIN_STATS build)               -> scan_shares.json
#      (traffic.json, valu.json, dynamic_mix.json and scan_shares.json are copied into profiles/<tag>/ ON THE BOX before the bench lines: bench.py quotes them by source hash)
#   3. rocprofv3 kernel trace + stats of the bench's timed steps, all four configurations             -> kernel_stats*.csv (must agree with kernel_ms)
#   4. the bench lines: headline (the driver's command line), the other BASELINE configs, budget, RCCL at world size 1, the closed loop
#   5. probes: seeds, small-batch latency, call latency, scheduler on / off, utilisation, scrub, cycles by section, timeline of a helped iteration
# usage (through gpurun):  bash scripts/profile_round.sh r05
set -u
TAG=${1:-r05}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT" "profiles/$TAG"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
V=$PWD/mpc_trajectory_generator_amd/csrc/variants
python -c "from mpc_trajectory_generator_amd import _lib; _lib.build_library(); _lib.build_variant(_lib.EXPERIMENTS)"      # (built before the call: they travel with the snapshot)
XLIB=$V/libnmpc_experiments.so
for f in ws:-DNMPC_WIN_STATS prof2:-DNMPC_PROF2 prof2e:-DNMPC_PROF2=2 tl:-DNMPC_TL; do
    [ -f $V/libnmpc_${f%%:*}.so ] || make -s -C mpc_trajectory_generator_amd/csrc -B OUT=variants/libnmpc_${f%%:*}.so EXTRA="${f#*:} -DNMPC_EXPERIMENTS"
done
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:$SQ1" "wait:$SQ2"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rocprofv3 --pmc $ctrs -d "$OUT/pmc_$name" -o pmc --output-format csv -- python scripts/pmc_one.py > "$OUT/pmc_$name.log" 2>&1
done
for pass in "lone_sq:$SQ1" "lone_wait:$SQ2"; do      # ONE wave on the chip: helpers off (a knob of the experiments build)
    name=${pass%%:*}; ctrs=${pass#*:}
    NMPC_LIB_PATH=$XLIB NMPC_TEAM_HELP=0 rocprofv3 --pmc $ctrs -d "$OUT/pmc_$name" -o pmc --output-format csv -- python scripts/pmc_lone.py > "$OUT/pmc_$name.log" 2>&1
done
rocprofv3 --pmc $SQ1 -d "$OUT/pmc_team_sq" -o pmc --output-format csv -- python scripts/pmc_lone.py > "$OUT/pmc_team_sq.log" 2>&1      # the same instance with its three helpers
rocprofv3 --pmc $SQ1 -d "$OUT/pmc_cfg2_sq" -o pmc --output-format csv -- python scripts/pmc_cfg2.py > "$OUT/pmc_cfg2_sq.log" 2>&1
# (scripts/bbcount.py run rebuilds its library when it was made from other sources)
{ timeout 600 python scripts/bbcount.py run hyb cfg1 8192; timeout 300 python scripts/bbcount.py run hyb cfg1 1 929; timeout 600 python scripts/bbcount.py run hyb cfg4 8192; } > "$OUT/bbcount.log" 2>&1
python scripts/bbcount.py collect "$OUT/dynamic_mix.json" gpurun_out/bbcount_hyb_cfg1.json gpurun_out/bbcount_hyb_cfg1_i929.json gpurun_out/bbcount_hyb_cfg4.json >> "$OUT/bbcount.log" 2>&1
cp "$OUT/dynamic_mix.json" "profiles/$TAG/" 2>/dev/null
for c in cfg1 cfg2 cfg3 cfg4; do python scripts/win_stats.py $c $V/libnmpc_ws.so; done > "$OUT/win_stats.txt" 2>&1
python scripts/profile_summarise.py "$OUT" > "$OUT/summarise.log" 2>&1
cp "$OUT/traffic.json" "$OUT/scan_shares.json" "$OUT/valu.json" "profiles/$TAG/" 2>/dev/null
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python bench.py --steps 5 --warmup 1 --no-extras > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
for c in cfg2 cfg3 cfg4; do
    rocprofv3 --kernel-trace --stats -d "$OUT/trace_$c" -o trace --output-format csv -- python bench.py --config $c --steps 3 --warmup 1 --no-extras > "$OUT/bench_${c}_under_rocprof.json" 2> "$OUT/trace_$c.log"
done
python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.log"
for c in cfg2 cfg3 cfg4; do python bench.py --config $c --steps 3 --warmup 1 --no-pipelined > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.log"; done
python bench.py --config cfg3 --batch 65536 --steps 2 --warmup 1 --no-extras > "$OUT/bench_cfg3_64k.json" 2> "$OUT/bench_cfg3_64k.log"
python bench.py --steps 5 --warmup 1 --budget 1500 --no-cpu-baseline > "$OUT/bench_budget1500.json" 2> "$OUT/bench_budget1500.log"
NMPC_BENCH_FORCE_DIST=1 python bench.py --steps 5 --warmup 1 --no-extras > "$OUT/bench_rccl_world1.json" 2> "$OUT/bench_rccl_world1.log"
python scripts/bench_receding.py > "$OUT/bench_receding.json" 2> "$OUT/bench_receding.log"
python scripts/perf_probe.py $TAG > "$OUT/perf_probe.json" 2> "$OUT/perf_probe.log"
for c in cfg1 cfg3 cfg4; do python scripts/seeds_unseen.py $c > "$OUT/seeds_unseen_$c.json" 2> "$OUT/seeds_unseen_$c.log"; done
for c in cfg1 cfg2; do python scripts/latency_team.py $c > "$OUT/latency_team_$c.json" 2> "$OUT/latency_team_$c.log"; done
for c in cfg1 cfg2 cfg3 cfg4; do python scripts/sched_ab.py $c "NMPC_SCHED=0" "NMPC_SCHED=1"; done > "$OUT/sched_ab.jsonl" 2> "$OUT/sched_ab.log"      # the step-aside scheduling off / on, seeds 0-2
for c in cfg1 cfg2 cfg3 cfg4; do python scripts/utilisation.py $c 0; done > "$OUT/utilisation.jsonl" 2> "$OUT/utilisation.log"      # busy wave-slot time without it
python scripts/call_latency.py > "$OUT/call_latency.txt" 2>&1
python scripts/scrub_probe.py shipped cfg1 cfg2 cfg3 cfg4 > "$OUT/scrub_probe.jsonl" 2>&1
{ NMPC_LIB_PATH=$V/libnmpc_prof2.so python scripts/sections.py cfg1 170 330; NMPC_TEAM_HELP=0 NMPC_LIB_PATH=$V/libnmpc_prof2.so python scripts/sections.py cfg1 170 330;
  NMPC_LIB_PATH=$V/libnmpc_prof2e.so python scripts/sections.py cfg1 170 330; } > "$OUT/sections.jsonl" 2> "$OUT/sections.log"
{ NMPC_LIB_PATH=$V/libnmpc_tl.so python scripts/timeline.py 170; NMPC_LIB_PATH=$V/libnmpc_tl.so python scripts/timeline.py 330; } > "$OUT/timeline.jsonl" 2> "$OUT/timeline.log"
python scripts/profile_summarise.py "$OUT" >> "$OUT/summarise.log" 2>&1
