"""One-stop timing probe used while tuning the hybrid kernel (not on the product path).

  headline   kernel ms of the cfg1 batch (B = 8192, 32 routes) for seeds 0, 1, 2
  lone       the two slowest instances of seed 0 solved alone: us per pass of a lone wave (the tail's speed)
  crowd      2048 copies of one hard instance (every wave slot busy with identical work): us per pass when two
             waves share each SIMD (the bulk's speed)
usage: python scripts/perf_probe.py [tag]        (NMPC_LIB_PATH selects an alternative build)"""
import json
import sys
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes

tag = sys.argv[1] if len(sys.argv) > 1 else "probe"
cfg = named_config("cfg1")
sol = BatchSolver(cfg, max_batch=8192)
out = {"tag": tag, "kernel": sol.kernel_name}
P0 = None
for seed in (0, 1, 2):
    P = synthetic_batch(cfg, 11, 8192, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed))
    if seed == 0:
        P0 = P
    sol.solve(P)
    ms = [(sol.solve(P), sol.last_batch_ms)[1] for _ in range(3)]
    st = sol.solve(P)[2]
    out[f"seed{seed}_ms"] = round(min(ms), 2)
    out[f"seed{seed}_maxpass"] = int(st["reserved"].max())
    if seed == 0:
        out["checksum"] = float(st["num_inner_iterations"].astype(np.float64).sum() + st["cost"].sum())
slowest = np.argsort(-sol.solve(P0)[2]["reserved"].astype(np.int64))[:2]
for b in (int(slowest[0]), int(slowest[1])):
    sol.solve(P0[b:b + 1])
    s = sol.solve(P0[b:b + 1])[2]
    out[f"lone{b}_passes"] = int(s["reserved"][0])
    out[f"lone{b}_us_per_pass"] = round(1e3 * sol.last_batch_ms / int(s["reserved"][0]), 3)
    out[f"lone{b}_ms"] = round(sol.last_batch_ms, 2)
Pc = np.repeat(P0[int(slowest[0]):int(slowest[0]) + 1], 2048, axis=0)
sol.solve(Pc)
s = sol.solve(Pc)[2]
out["crowd_us_per_pass"] = round(1e3 * sol.last_batch_ms / int(s["reserved"][0]), 3)
print(json.dumps(out))
