"""Boils the rocprofv3 output of scripts/profile_round.sh down to the small CSVs kept under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
# kernel stats: rocprofv3 writes <dir>/<host>/<pid>_kernel_stats.csv or <dir>/trace_kernel_stats.csv
cands = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if cands:
    rows = list(csv.reader(open(cands[0])))
    with open(os.path.join(out, "kernel_stats.csv"), "w", newline="") as fh:
        csv.writer(fh).writerows(rows)
for name in ("fetch", "write", "sq", "wait"):
    files = glob.glob(os.path.join(out, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = defaultdict(lambda: [0.0, 0, None])
    for r in csv.DictReader(open(files[0])):
        key = (r["Kernel_Name"], r["Counter_Name"])
        agg[key][0] += float(r["Counter_Value"])
        agg[key][1] += 1
        agg[key][2] = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
    with open(os.path.join(out, f"pmc_{name}.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size", "Counter_Name",
                    "Dispatches", "Counter_Value_Per_Dispatch"])
        for (k, c), (tot, n, meta) in sorted(agg.items()):
            w.writerow([k, *meta, c, n, tot / n])
    log = os.path.join(out, f"pmc_{name}.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("launches"):
                with open(os.path.join(out, f"pmc_{name}.csv"), "a") as fh:
                    fh.write("# " + line)
print(open(os.path.join(out, "bench.json")).read().strip()[:400])
