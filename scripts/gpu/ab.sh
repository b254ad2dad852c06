# A/B of two builds of the library on ONE box: scripts/gpu/ab.sh <tag> <base.so> [pytest]
# (perf_probe for both, twice, alternating; then the quick bit-exactness check and -- with a third argument -- pytest -m gpu on the new build)
set -u
TAG=$1; BASE=$PWD/$2; OUT=gpurun_out/$TAG; mkdir -p $OUT
python scripts/gpu_check.py > $OUT/gpu_check.log 2>&1; tail -2 $OUT/gpu_check.log
for r in 1 2; do
  NMPC_LIB_PATH=$BASE python scripts/perf_probe.py base$r > $OUT/probe_base$r.json 2> $OUT/probe_base$r.err; cat $OUT/probe_base$r.json
  python scripts/perf_probe.py new$r > $OUT/probe_new$r.json 2> $OUT/probe_new$r.err; cat $OUT/probe_new$r.json
done
if [ $# -ge 3 ]; then timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log; fi
