set -u
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
O=gpurun_out/r06_g8; mkdir -p $O
timeout 600 python scripts/bbcount.py run hyb cfg1 8192 > $O/bb_hyb.log 2>&1; echo "bb hyb rc=$?"; tail -1 $O/bb_hyb.log
