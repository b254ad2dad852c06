"""Kernel ms of the headline batch (seeds 0, 1, 2) for the migration knobs given in the environment."""
import os, sys, json
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
cfg = named_config("cfg1")
sol = BatchSolver(cfg, max_batch=8192)
out = {"min": os.environ.get("NMPC_PARK_MIN"), "depth": os.environ.get("NMPC_PARK_DEPTH")}
for seed in (0, 1, 2):
    P = synthetic_batch(cfg, 11, 8192, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed))
    sol.solve(P)
    out[f"s{seed}"] = round(min((sol.solve(P), sol.last_batch_ms)[1] for _ in range(3)), 1)
out["mean"] = round((out["s0"] + out["s1"] + out["s2"]) / 3, 1)
print(json.dumps(out))
