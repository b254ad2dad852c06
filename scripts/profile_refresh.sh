#!/bin/bash
# The part of scripts/profile_round.sh that is keyed by the hash of the kernel sources: kernel trace + stats of the bench's timed steps, the
# FETCH_SIZE / WRITE_SIZE passes (traffic.json), and the headline bench line.  For a change that leaves the kernels' machine code alone
# (a host-side knob, a header comment) -- the rest of profiles/<tag>/ stays as collected.   usage (through gpurun): bash scripts/profile_refresh.sh r04
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_${TAG}_refresh
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python bench.py --steps 5 --warmup 1 --no-extras > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rocprofv3 --pmc $ctrs -d "$OUT/pmc_$name" -o pmc --output-format csv -- python scripts/pmc_one.py > "$OUT/pmc_$name.log" 2>&1
done
python scripts/profile_summarise.py "$OUT" > "$OUT/summarise.log" 2>&1      # writes traffic.json (the bench line below then quotes it if it is copied to profiles/)
mkdir -p profiles/$TAG && cp "$OUT/traffic.json" profiles/$TAG/traffic.json
python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.log"
