set -u
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
O=gpurun_out/r06_g9; mkdir -p $O
timeout 300 python scripts/perf_probe.py cwlds > $O/perf_probe.json 2> $O/perf_probe.log; echo "probe rc=$?"; cat $O/perf_probe.json
for c in cfg2; do timeout 600 python bench.py --config $c --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.log; echo "bench $c rc=$? $(cut -c1-140 $O/bench_$c.json)"; done
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
