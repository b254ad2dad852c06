"""One hard instance solved alone (a single wavefront on the chip), twice: the workload of lone-wave PMC runs."""
import sys
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
cfg = named_config("cfg1")
sol = BatchSolver(cfg, max_batch=8192)
P = synthetic_batch(cfg, 11, 8192, 0, routes=random_routes(cfg, 11, 32, seed=1000))
b = int(sys.argv[1]) if len(sys.argv) > 1 else 170
for _ in range(2):
    u, y, st = sol.solve(P[b:b + 1])
print("launches 2 passes/launch", int(st["reserved"][0]), "iters/launch", int(st["num_inner_iterations"][0]), "ms", sol.last_batch_ms)
