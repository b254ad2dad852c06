for v in cur_maxilp_w1 cur_maxilp_noaa cur_maxilp_nosb; do
  NMPC_LIB_PATH=scripts/variants/$v.so timeout 900 python scripts/hyb2_caps.py $v 48 
done > gpurun_out/hyb2_caps.jsonl 2> gpurun_out/hyb2_caps.err
cat gpurun_out/hyb2_caps.jsonl; tail -5 gpurun_out/hyb2_caps.err
