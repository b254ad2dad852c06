"""experiments: operands of the first Lipschitz tests of instance 0 (library built with -DNMPC2_DEBUG_LIP); the last instance's y_out row is the debug area"""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
cfg = named_config("cfg2")
P = synthetic_batch(cfg, 11, 2, 0, routes=random_routes(cfg, 11, 32, seed=1000))
P[1] = P[0]
os.environ["NMPC_TEAM_HELP"] = "0"; os.environ["NMPC_TEAM_OWNERS"] = "1"
s = BatchSolver(cfg, max_batch=2, max_outer=1, max_inner=1)
u, y, st = s.solve(P)
print("n_cost", st["num_cost_evals"], "n_grad", st["num_grad_evals"])
for k in range(4):
    print(k, " ".join("%s=%.17g" % (n, v) for n, v in zip(("cost", "gr", "nr2", "c_lip", "psiA", "rhs", "gamma", "Lc"), y[1][8 * k: 8 * k + 8])))
