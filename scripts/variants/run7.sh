for v in iilp_dbg mm_dbg; do echo "== $v"; NMPC_LIB_PATH=scripts/variants/$v.so python scripts/hyb2_dbg.py 2>&1 | cut -c1-400; done | tee gpurun_out/hyb2_dbg.txt
