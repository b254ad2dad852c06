v=new_maxmem
for caps in "1 1" "1 3" "1 30" "2 30"; do echo "== $v caps $caps"; NMPC_LIB_PATH=scripts/variants/$v.so python scripts/hyb2_caps2.py $caps 6; done 2>&1 | tee gpurun_out/hyb2_caps3.txt
