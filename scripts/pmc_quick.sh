#!/bin/bash
set -u
OUT=$PWD/gpurun_out/pmc_quick; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
rocprofv3 --pmc $SQ1 -d "$OUT/pmc_sq" -o pmc --output-format csv -- python scripts/pmc_one.py > "$OUT/pmc_sq.log" 2>&1
NMPC_LIB_PATH=$PWD/mpc_trajectory_generator_amd/csrc/variants/libnmpc_experiments.so NMPC_TEAM_HELP=0 rocprofv3 --pmc $SQ1 -d "$OUT/pmc_lone_sq" -o pmc --output-format csv -- python scripts/pmc_lone.py > "$OUT/pmc_lone_sq.log" 2>&1
python - <<'PY'
import csv, glob, collections
for name in ("pmc_sq","pmc_lone_sq"):
    tot=collections.defaultdict(float); n=0
    for f in glob.glob(f"gpurun_out/pmc_quick/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "solve_hyb" in r["Kernel_Name"]:
                tot[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(name, dict(tot))
PY
tail -3 $OUT/pmc_sq.log; tail -3 $OUT/pmc_lone_sq.log
