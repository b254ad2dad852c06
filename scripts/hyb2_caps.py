"""Where does a build of nmpc_solve_hyb2_kernel first leave the oracle?  Config-2 instances under tiny iteration caps (max_outer max_inner n [opt=value ...]): per-instance counters and reals, HIP (the library in NMPC_LIB_PATH) against the oracle.  The max-ilp build of round 3 was wrong from the first line search on (n_grad 14 instead of 4: every trial rejected), which pointed at pair_sum."""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import oracle_for
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
mo, mi, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
extra = dict(kv.split("=") for kv in sys.argv[4:])
extra = {k: int(v) for k, v in extra.items()}
cfg = named_config("cfg2")
P = synthetic_batch(cfg, 11, n, 0, routes=random_routes(cfg, 11, 32, seed=1000))
os.environ["NMPC_TEAM_HELP"] = "0"
s = BatchSolver(cfg, max_batch=n, max_outer=mo, max_inner=mi, **extra)
u, y, st = s.solve(P)
uo, yo, sto = oracle_for(cfg, **s.oracle_opts()).solve_batch(P, threads=16)
for i in range(n):
    print(i, "gpu", [int(st[f][i]) for f in ("num_inner_iterations", "num_cost_evals", "num_grad_evals", "reserved")], "%.17g %.6g" % (st["cost"][i], st["last_problem_norm_fpr"][i]),
          "| orc", [int(sto[f][i]) for f in ("num_inner_iterations", "num_cost_evals", "num_grad_evals")], "%.17g %.6g" % (sto["cost"][i], sto["last_problem_norm_fpr"][i]),
          "| du %.3g" % np.abs(u[i] - uo[i]).max(), "first bad stage", int(np.argmax(u[i] != uo[i])) if (u[i] != uo[i]).any() else -1)
