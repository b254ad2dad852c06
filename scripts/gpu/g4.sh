set -u
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
O=gpurun_out/r06_g4; mkdir -p $O
timeout 300 python scripts/perf_probe.py execwin > $O/perf_probe.json 2> $O/perf_probe.log; echo "probe rc=$?"; cat $O/perf_probe.json
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python scripts/bbcount.py run hyb cfg1 8192 > $O/bb_full.log 2>&1; echo "bb full rc=$?"; tail -1 $O/bb_full.log
timeout 600 python bench.py --config cfg2 --steps 3 --warmup 1 --no-extras > $O/bench_cfg2.json 2> $O/bench_cfg2.log; echo "bench cfg2 rc=$?"; cut -c1-300 $O/bench_cfg2.json
timeout 600 python bench.py --config cfg3 --steps 3 --warmup 1 --no-extras > $O/bench_cfg3.json 2> $O/bench_cfg3.log; echo "bench cfg3 rc=$?"; cut -c1-300 $O/bench_cfg3.json
