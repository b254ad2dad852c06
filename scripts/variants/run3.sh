for v in cur_maxilp_w1; do
  echo "== $v caps 1 1"; NMPC_LIB_PATH=scripts/variants/$v.so python scripts/hyb2_caps2.py 1 1 8
  echo "== $v caps 1 2"; NMPC_LIB_PATH=scripts/variants/$v.so python scripts/hyb2_caps2.py 1 2 8
  echo "== $v caps 1 1 akkt=2"; NMPC_LIB_PATH=scripts/variants/$v.so python scripts/hyb2_caps2.py 1 1 8 akkt_gradient=2
done 2>&1 | tee gpurun_out/hyb2_caps2.txt
