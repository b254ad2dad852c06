"""Helpers for the K longest instances of a full batch BEFORE the queue is dry (KArgs.topk, nmpc_solve_hyb.h): kernel ms per seed and a
checksum of the results (which must not move: only where work runs changes) for K in a list, experiments build.
usage: python scripts/topk_ab.py cfgN [K,K,...] [min_pass,...]   -> one JSON line per (K, min_pass)"""
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
Ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,16,64").split(",")]
Ts = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "3000").split(",")]
cfg = named_config(name)
kw = dict(synthetic_circles=(name == "cfg3"), random_dyn=(name == "cfg4"))
Ps = [synthetic_batch(cfg, 11, 8192, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed), **kw) for seed in (0, 1, 2)]
for T in Ts:
    for K in Ks:
        os.environ["NMPC_TOPK"], os.environ["NMPC_TOPK_PASS"] = str(K), str(T)
        sol = BatchSolver(cfg, max_batch=8192, experiments=True)
        row = {"config": name, "topk": K, "min_pass": T, "ms": [], "checksum": []}
        for P in Ps:
            sol.solve(P)
            ms = []
            for _ in range(3):
                u, y, st = sol.solve(P)
                ms.append(sol.last_batch_ms)
            row["ms"].append(round(min(ms), 2))
            row["checksum"].append(float(st["num_inner_iterations"].astype(np.float64).sum() + st["cost"].sum() + u.sum()))
        row["mean_ms"] = round(float(np.mean(row["ms"])), 2)
        print(json.dumps(row), flush=True)
        sol.close()
        if K == 0:
            break_after_first_T = True
