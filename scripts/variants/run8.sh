python scripts/scrub_probe.py shipped cfg2 cfg1 cfg3 cfg4
for v in new_maxmem new_maxilp new_default; do NMPC_LIB_PATH=scripts/variants/$v.so python scripts/scrub_probe.py $v cfg2; done
