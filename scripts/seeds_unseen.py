import sys, json
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
cfg = named_config("cfg1")
sol = BatchSolver(cfg, max_batch=8192)
out = {}
for seed in (3, 4, 5, 6, 7):
    P = synthetic_batch(cfg, 11, 8192, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed))
    sol.solve(P)
    r = [sol.solve(P)[2] for _ in range(3)]
    ms = [float(s["solve_time_ms"][0]) for s in r]
    st = r[0]
    out[seed] = {"ms_mean": round(float(np.mean(ms)), 1), "ms_min": round(min(ms), 1), "max_passes": int(st["reserved"].max()),
                 "floor_ms_at_5.2us": round(5.2e-3 * int(st["reserved"].max()), 1), "mean_passes": round(float(st["reserved"].mean()))}
print(json.dumps(out))
