set -u
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
O=gpurun_out/r06_g11; mkdir -p $O
timeout 400 python scripts/bbcount.py run hyb2 cfg2 192 > $O/bb_hyb2.log 2>&1; echo "bb hyb2 rc=$?"; tail -2 $O/bb_hyb2.log
timeout 600 python -m pytest tests/test_bbcount.py -m gpu -q > $O/pytest_bb.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_bb.log
