set -u
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
O=gpurun_out/r06_g3; mkdir -p $O
timeout 600 python scripts/excl_sweep.py cfg1 "0" "1000,256,0" "2000,256,0" "3000,256,0" "4000,256,0" "2000,128,0" "2000,512,0" "3000,512,0" "2000,256,1" "3000,256,1" "0" > $O/excl_cfg1.jsonl 2> $O/excl_cfg1.log; echo "excl cfg1 rc=$?"; cat $O/excl_cfg1.jsonl | cut -c1-200
timeout 600 python scripts/excl_sweep.py cfg3 "0" "2000,256,0" "3000,256,0" "3000,512,0" "0" > $O/excl_cfg3.jsonl 2> $O/excl_cfg3.log; echo "excl cfg3 rc=$?"; cat $O/excl_cfg3.jsonl | cut -c1-200
timeout 600 python scripts/excl_sweep.py cfg4 "0" "2000,256,0" "3000,256,0" "3000,512,0" "0" > $O/excl_cfg4.jsonl 2> $O/excl_cfg4.log; echo "excl cfg4 rc=$?"; cat $O/excl_cfg4.jsonl | cut -c1-200
for v in 1 0 1 0; do NMPC_LOOP_ORDER_PREV=$v timeout 300 python scripts/bench_receding.py --experiments > $O/receding_prev$v.json 2>> $O/receding.log; echo "receding prev=$v: $(cut -c1-250 $O/receding_prev$v.json)"; done
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
