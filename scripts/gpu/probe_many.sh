# perf_probe for several builds on ONE box, twice round-robin: scripts/gpu/probe_many.sh <tag> <lib.so> ...   ("shipped" = the in-tree library)
set -u
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
for r in 1 2; do
  for lib in "$@"; do
    n=$(basename $lib .so)
    if [ "$lib" = shipped ]; then python scripts/perf_probe.py shipped_$r; else NMPC_LIB_PATH=$PWD/$lib python scripts/perf_probe.py ${n}_$r; fi 2>> $OUT/err.log | tee -a $OUT/probes.jsonl
  done
done
