#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline numbers on the GPU box:
#   1. kernel trace + stats of the default bench run        -> kernel_stats.csv
#   2. PMC passes, each in its own run (no trace domains):  FETCH_SIZE | WRITE_SIZE | SQ instruction mix | SQ waits
# usage (through gpurun):  bash scripts/profile_round.sh r01      ; results under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pipelined > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" \
            "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
            "wait:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rocprofv3 --pmc $ctrs -d "$OUT/pmc_$name" -o pmc --output-format csv -- python scripts/pmc_one.py > "$OUT/pmc_$name.log" 2>&1
done
python bench.py --steps 10 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.log"
python scripts/profile_summarise.py "$OUT"
