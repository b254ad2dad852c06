set -x
python scripts/hyb2_repro.py ref > gpurun_out/hyb2_repro.jsonl 2> gpurun_out/hyb2_repro.err
for v in cur_iilp_w2 cur_maxilp_w1 cur_default_w2 cur_maxilp_w2 b86_iilp_w1 b86_iilp_w2 b86_maxilp_w1; do
  NMPC_LIB_PATH=scripts/variants/$v.so timeout 600 python scripts/hyb2_repro.py probe $v >> gpurun_out/hyb2_repro.jsonl 2>> gpurun_out/hyb2_repro.err
done
cat gpurun_out/hyb2_repro.jsonl | cut -c1-400
