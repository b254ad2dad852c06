python scripts/hyb2_repro.py ref > gpurun_out/hyb2_repro2.jsonl 2> gpurun_out/hyb2_repro2.err
for v in new_maxilp new_default new_maxmem new_iilp_w2 new_maxilp_w2; do
  NMPC_LIB_PATH=scripts/variants/$v.so timeout 600 python scripts/hyb2_repro.py probe $v >> gpurun_out/hyb2_repro2.jsonl 2>> gpurun_out/hyb2_repro2.err
done
grep -v '"alone"' gpurun_out/hyb2_repro2.jsonl | cut -c1-260
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
