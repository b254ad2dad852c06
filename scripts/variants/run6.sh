for v in mm_nopost mm_noaa mm_nosb mm_noagpr; do echo "== $v"; NMPC_LIB_PATH=scripts/variants/$v.so python scripts/hyb2_caps2.py 1 1 3 2>&1 | cut -c1-120; done | tee gpurun_out/hyb2_caps4.txt
