"""A/B of the SIMD-exclusive long instances (KArgs.excl, nmpc_solve_hyb.h) on one box: kernel ms of a config's batch for a grid of
(NMPC_EXCL_MIN, NMPC_EXCL_CAP, NMPC_EXCL_YIELD) settings of the experiments build, same checksum demanded in every row.
usage: python profiles/r06/simd_exclusive_sweep.py (with profiles/r06/simd_exclusive.patch applied) cfg1 "0" "2000,256,0" "3000,128,1" ..."""
import json
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes

name = sys.argv[1]
cfg = named_config(name)
B = 8192
batches = [synthetic_batch(cfg, 11, B, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed), synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
           for seed in (0, 1)]
ref = None
for setting in sys.argv[2:]:
    v = [int(x) for x in setting.split(",")] + [0, 0]
    os.environ["NMPC_EXCL_MIN"], os.environ["NMPC_EXCL_CAP"], os.environ["NMPC_EXCL_YIELD"] = str(v[0]), str(v[1] or 256), str(v[2])
    s = BatchSolver(cfg, max_batch=B, experiments=True)
    row = {"cfg": name, "excl_min": v[0], "excl_cap": v[1] or 256, "yield": v[2]}
    sums = []
    for k, P in enumerate(batches):
        s.solve(P)
        ms = []
        for _ in range(3):
            u, y, st = s.solve(P)
            ms.append(s.last_batch_ms)
        row[f"seed{k}_ms"] = round(min(ms), 2)
        sums.append(float(st["num_inner_iterations"].astype(np.float64).sum() + st["cost"].sum() + np.abs(u).sum()))
    row["checksum"] = sums
    if ref is None:
        ref = sums
    row["same_bits_as_first_row"] = sums == ref
    s.close()
    print(json.dumps(row), flush=True)
