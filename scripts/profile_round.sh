#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline numbers on the GPU box:
#   1. kernel trace + stats of the bench's timed steps          -> kernel_stats.csv   (must agree with bench's kernel_ms)
#   2. PMC passes, each in its own run (no trace domains): FETCH_SIZE | WRITE_SIZE | SQ instruction mix | SQ waits
#                                                              -> pmc_*.csv, traffic.json (what bench.py quotes as `traffic`)
#   3. the same SQ passes for ONE hard instance solved alone (a single wavefront): what a pass costs without a neighbour
#   4. the bench lines: headline, the other BASELINE configs, the iteration-budget line, the receding-horizon loop
# usage (through gpurun):  bash scripts/profile_round.sh r02      ; results under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python bench.py --steps 5 --warmup 1 --no-extras > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:$SQ1" "wait:$SQ2"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rocprofv3 --pmc $ctrs -d "$OUT/pmc_$name" -o pmc --output-format csv -- python scripts/pmc_one.py > "$OUT/pmc_$name.log" 2>&1
done
python -c "from mpc_trajectory_generator_amd import _lib; _lib.build_variant(_lib.EXPERIMENTS)"      # the knobs (NMPC_TEAM_HELP, ...) exist in this variant only
XLIB=$PWD/mpc_trajectory_generator_amd/csrc/variants/libnmpc_experiments.so
for pass in "lone_sq:$SQ1" "lone_wait:$SQ2"; do      # ONE wave on the chip: helpers off
    name=${pass%%:*}; ctrs=${pass#*:}
    NMPC_LIB_PATH=$XLIB NMPC_TEAM_HELP=0 rocprofv3 --pmc $ctrs -d "$OUT/pmc_$name" -o pmc --output-format csv -- python scripts/pmc_lone.py > "$OUT/pmc_$name.log" 2>&1
done
rocprofv3 --pmc $SQ1 -d "$OUT/pmc_team_sq" -o pmc --output-format csv -- python scripts/pmc_lone.py > "$OUT/pmc_team_sq.log" 2>&1      # the same instance with its three helpers
python bench.py --steps 10 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.log"
for c in cfg2 cfg3 cfg4; do python bench.py --config $c --steps 3 --warmup 1 --no-pipelined > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.log"; done
for c in cfg2 cfg3 cfg4; do      # the same kernel trace for the other BASELINE configurations
    rocprofv3 --kernel-trace --stats -d "$OUT/trace_$c" -o trace --output-format csv -- python bench.py --config $c --steps 3 --warmup 1 --no-extras > "$OUT/bench_${c}_under_rocprof.json" 2> "$OUT/trace_$c.log"
done
python bench.py --config cfg3 --batch 65536 --steps 2 --warmup 1 --no-extras > "$OUT/bench_cfg3_64k.json" 2> "$OUT/bench_cfg3_64k.log"
python bench.py --steps 5 --warmup 1 --budget 1500 --no-cpu-baseline > "$OUT/bench_budget1500.json" 2> "$OUT/bench_budget1500.log"
NMPC_BENCH_FORCE_DIST=1 python bench.py --steps 5 --warmup 1 --no-extras > "$OUT/bench_rccl_world1.json" 2> "$OUT/bench_rccl_world1.log"
python scripts/bench_receding.py > "$OUT/bench_receding.json" 2> "$OUT/bench_receding.log"
python scripts/perf_probe.py $TAG > "$OUT/perf_probe.json" 2> "$OUT/perf_probe.log"
for c in cfg1 cfg3 cfg4; do python scripts/seeds_unseen.py $c > "$OUT/seeds_unseen_$c.json" 2> "$OUT/seeds_unseen_$c.log"; done
for c in cfg1 cfg2; do python scripts/latency_team.py $c > "$OUT/latency_team_$c.json" 2> "$OUT/latency_team_$c.log"; done
rocprofv3 --pmc $SQ1 -d "$OUT/pmc_cfg2_sq" -o pmc --output-format csv -- python scripts/pmc_cfg2.py > "$OUT/pmc_cfg2_sq.log" 2>&1
for c in cfg1 cfg2 cfg3 cfg4; do python scripts/sched_ab.py $c "NMPC_SCHED=0" "NMPC_SCHED=1"; done > "$OUT/sched_ab.jsonl" 2> "$OUT/sched_ab.log"      # the step-aside scheduling off / on, seeds 0-2
for c in cfg1 cfg2 cfg3 cfg4; do python scripts/utilisation.py $c 0; done > "$OUT/utilisation.jsonl" 2> "$OUT/utilisation.log"      # busy wave-slot time without it
python scripts/call_latency.py > "$OUT/call_latency.txt" 2>&1
python scripts/scrub_probe.py shipped cfg1 cfg2 cfg3 cfg4 > "$OUT/scrub_probe.jsonl" 2>&1
python scripts/profile_summarise.py "$OUT"
