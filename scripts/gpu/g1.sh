set -u
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
O=gpurun_out/r06_g1; mkdir -p $O
timeout 300 python scripts/bbcount.py run hyb cfg1 256 > $O/bb_small.log 2>&1; echo "bb small rc=$?"
timeout 600 python scripts/bbcount.py run hyb cfg1 8192 > $O/bb_full.log 2>&1; echo "bb full rc=$?"
timeout 300 python scripts/bbcount.py run hyb cfg1 1 170 > $O/bb_lone.log 2>&1; echo "bb lone rc=$?"
timeout 600 python scripts/bbcount.py run hyb cfg4 8192 > $O/bb_cfg4.log 2>&1; echo "bb cfg4 rc=$?"
timeout 300 python scripts/perf_probe.py base > $O/perf_probe.json 2> $O/perf_probe.log; echo "probe rc=$?"
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1000 -d $O/pcs -o pcs --output-format csv -- python scripts/pmc_one.py > $O/pcs.log 2>&1; echo "pcs rc=$?"
ls -la $O/pcs 2>/dev/null | head; du -sh $O
tail -3 $O/*.log
