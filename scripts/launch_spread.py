"""Kernel ms of N consecutive launches of a config's batch (the spread a multi-step mean carries).  usage: python scripts/launch_spread.py [cfgN] [n]"""
import sys, json, numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = named_config(name)
P = synthetic_batch(cfg, 11, 8192, 0, routes=random_routes(cfg, 11, 32, seed=1000), synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
sol = BatchSolver(cfg, max_batch=8192)
sol.solve(P)
ms, slow = [], []
for _ in range(n):
    st = sol.solve(P)[2]; ms.append(round(sol.last_batch_ms, 2)); slow.append(round(float(st["solve_time_ms"].max()), 1))
print(json.dumps({"ms": ms, "slowest_instance_ms": slow, "mean": round(float(np.mean(ms)), 2), "min": min(ms), "max": max(ms)}))
