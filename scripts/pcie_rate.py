"""Host-buffer path (nmpc_solve_batch_host: H2D copy of p/u/y, solve, D2H copy of u/y/status) on the
headline batch: the PCIe-inclusive rate quoted in DESIGN.md section 6 (never bench.py's `value`)."""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
cfg = named_config("cfg1")
sol = BatchSolver(cfg, max_batch=8192)
P = synthetic_batch(cfg, 11, 8192, 0, routes=random_routes(cfg, 11, 32, seed=1000))
sol.solve(P)
ts = []
for _ in range(5):
    t = time.perf_counter()
    u, y, st = sol.solve(P)
    ts.append(time.perf_counter() - t)
print(f"host path: {1e3 * np.mean(ts):.2f} ms per 8192-batch incl. PCIe both ways = {8192 / np.mean(ts):.0f} solves/s; "
      f"kernel alone {sol.last_batch_ms:.2f} ms")
