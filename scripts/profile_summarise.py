"""Boils the rocprofv3 output of scripts/profile_round.sh down to the small files kept under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, ".")
out = sys.argv[1]
# kernel stats: rocprofv3 writes <dir>/<host>/<pid>_kernel_stats.csv or <dir>/trace_kernel_stats.csv
cands = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if cands:
    rows = list(csv.reader(open(cands[0])))
    with open(os.path.join(out, "kernel_stats.csv"), "w", newline="") as fh:
        csv.writer(fh).writerows(rows)
for c in ("cfg2", "cfg3", "cfg4"):
    cands = glob.glob(os.path.join(out, f"trace_{c}", "**", "*kernel_stats.csv"), recursive=True)
    if cands:
        with open(os.path.join(out, f"kernel_stats_{c}.csv"), "w", newline="") as fh:
            csv.writer(fh).writerows(list(csv.reader(open(cands[0]))))
per_kernel = {}
for name in ("fetch", "write", "sq", "wait", "lone_sq", "lone_wait", "team_sq", "cfg2_sq"):
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
            per_kernel[(name, k, c)] = tot / n
    log = os.path.join(out, f"pmc_{name}.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("launches"):
                with open(os.path.join(out, f"pmc_{name}.csv"), "a") as fh:
                    fh.write("# " + line)
# traffic.json: HBM bytes per launch of the solve kernel = FETCH_SIZE (KiB) x 2 [gfx950: wide reads are tallied at half
# their bytes, MI355X_MICROARCH.md section HBM] + WRITE_SIZE (KiB), keyed by kernel + source hash + workload
from mpc_trajectory_generator_amd import _lib      # noqa: E402
fetch = {k: v for (n, k, c), v in per_kernel.items() if n == "fetch" and c == "FETCH_SIZE" and "solve" in k}
write = {k: v for (n, k, c), v in per_kernel.items() if n == "write" and c == "WRITE_SIZE" and "solve" in k}
entries = []
for k in fetch:
    if k in write:
        short = k.split("(")[0].replace("void ", "").replace("nmpc::", "")
        entries.append({"kernel": short.replace("<nmpc::", "<").strip(), "source_hash": _lib.source_hash(), "config": "cfg1", "batch": 8192,
                        "routes": 32, "fetch_size_kib": fetch[k], "write_size_kib": write[k],
                        "hbm_bytes_per_launch": 2 * fetch[k] * 1024 + write[k] * 1024,
                        "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of scripts/pmc_one.py (two launches of the "
                               "headline batch); FETCH_SIZE x 2 and WRITE_SIZE x 1: calibrated on this kernel's own 8 B/lane access width, profiles/r04/fetch_calibration.json"})
with open(os.path.join(out, "traffic.json"), "w") as fh:
    json.dump(entries, fh, indent=1)
print(json.dumps(entries))
# valu.json: instructions per evaluation pass and how busy the vector ALU is, from the SQ counter pass of the same batch (pmc_sq) and of the cfg 2
# batch (pmc_cfg2_sq), keyed like the traffic -- bench.py quotes roofline.valu_per_pass / roofline.valu_busy from it
valu = []
for name, cfgname, waves_per_simd in (("sq", "cfg1", 2), ("cfg2_sq", "cfg2", 1)):
    log = os.path.join(out, f"pmc_{name}.log")
    passes = None
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("launches"):
                w_ = line.split()
                if "passes/launch" in w_:
                    passes = float(w_[w_.index("passes/launch") + 1])
    for k in sorted({k for (n, k, c) in per_kernel if n == name and "solve" in k}):
        g = lambda c: per_kernel.get((name, k, c))      # noqa: E731
        if passes and g("SQ_INSTS_VALU") and g("SQ_WAVE_CYCLES"):
            short = k.split("(")[0].replace("void ", "").replace("nmpc::", "").replace("<nmpc::", "<").strip()
            valu.append({"kernel": short, "source_hash": _lib.source_hash(), "config": cfgname, "batch": 8192, "passes_per_launch": passes,
                         "valu_per_pass": g("SQ_INSTS_VALU") / passes, "salu_per_pass": (g("SQ_INSTS_SALU") or 0.0) / passes,
                         "lds_per_pass": (g("SQ_INSTS_LDS") or 0.0) / passes,
                         "valu_busy_per_wave": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), "waves_per_simd": waves_per_simd,
                         "valu_busy": waves_per_simd * g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"),
                         "how": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES (one run, no trace domains) of two launches of "
                                "the batch; per launch / evaluation passes of the launch (sum of nmpc_status.reserved); valu_busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x resident waves per SIMD"})
with open(os.path.join(out, "valu.json"), "w") as fh:
    json.dump(valu, fh, indent=1)
print(json.dumps(valu))
# scan_shares.json: share of the evaluations in which the exact certificates of eval_psi fell back (scripts/win_stats.py on a -DNMPC_WIN_STATS
# build), keyed like the traffic -- bench.py derives roofline.executed_frac from it
ws = os.path.join(out, "win_stats.txt")
if os.path.exists(ws):
    shares = []
    for line in open(ws):
        i = line.find("{")
        if i >= 0:
            try:
                d = json.loads(line[i:])
            except ValueError:
                continue
            shares.append({k: d.get(k) for k in ("kernel", "source_hash", "config", "searches", "fallbacks", "window_full_scan_share",
                                                 "obstacle_certificates", "obstacle_scans", "obstacle_scan_share")})
    with open(os.path.join(out, "scan_shares.json"), "w") as fh:
        json.dump(shares, fh, indent=1)
    print(json.dumps(shares))
b = os.path.join(out, "bench.json")
if os.path.exists(b):
    print(open(b).read().strip()[:400])
