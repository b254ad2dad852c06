set -u
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
O=gpurun_out/r06_g5; mkdir -p $O
timeout 600 python scripts/bbcount.py run hyb cfg1 8192 > $O/bb_hyb.log 2>&1; echo "bb hyb rc=$?"; tail -1 $O/bb_hyb.log
timeout 600 python scripts/bbcount.py run hyb cfg1 1 929 > $O/bb_hyb_lone.log 2>&1; echo "bb lone rc=$?"; tail -1 $O/bb_hyb_lone.log
timeout 900 python scripts/bbcount.py run hyb2 cfg2 8192 > $O/bb_hyb2.log 2>&1; echo "bb hyb2 rc=$?"; tail -1 $O/bb_hyb2.log
timeout 300 python scripts/perf_probe.py base2 > $O/perf_probe.json 2> $O/perf_probe.log; echo "probe rc=$?"; cat $O/perf_probe.json
