import sys, json
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
cfg = named_config(cfg_name)
kw = dict(synthetic_circles=cfg_name == "cfg3", random_dyn=cfg_name == "cfg4")
sol = BatchSolver(cfg, max_batch=8192)
out = {"config": cfg_name}
for seed in (3, 4, 5, 6, 7):
    P = synthetic_batch(cfg, 11, 8192, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed), **kw)
    sol.solve(P)
    ms = []
    for _ in range(3):
        st = sol.solve(P)[2]
        ms.append(sol.last_batch_ms)
    out[seed] = {"ms_mean": round(float(np.mean(ms)), 1), "ms_min": round(min(ms), 1), "max_passes": int(st["reserved"].max()),
                 "floor_ms_at_5.2us": round(5.2e-3 * int(st["reserved"].max()), 1), "mean_passes": round(float(st["reserved"].mean()))}
print(json.dumps(out))
