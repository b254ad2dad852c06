import sys
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
cfg = named_config("cfg1")
sol = BatchSolver(cfg, max_batch=8192)
for seed in (0, 1):
    P = synthetic_batch(cfg, 11, 8192, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed))
    u, y, st = sol.solve(P)
    np.savez_compressed(f"gpurun_out/passes_seed{seed}.npz", passes=st["reserved"], iters=st["num_inner_iterations"], outer=st["num_outer_iterations"],
                        exit=st["exit_status"], penalty=st["penalty"], f2=st["f2_norm"], dy=st["delta_y_norm_over_c"])
