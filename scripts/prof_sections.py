"""Per-section cycle counts of single instances (library built with -DNMPC_PROFILE).
The status fields are overloaded with s_memtime deltas by that build."""
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
for b in (170, 4175, 0, 1):
    sol.solve(P[b:b + 1])
    _, _, s = sol.solve(P[b:b + 1])
    n = s["reserved"][0]
    print(f"inst {b}: passes {n} | cycles/pass: eval {s['last_problem_norm_fpr'][0]/n:.0f} "
          f"[rollout {s['penalty'][0]/n:.0f} stage+CTE {s['cost'][0]/n:.0f} acc+sum {s['solve_time_ms'][0]/n:.0f} circles {100.0*s['num_outer_iterations'][0]/n:.0f} "
          f"ellipses {100.0*s['num_inner_iterations'][0]/n:.0f} F2sums {100.0*s['num_cost_evals'][0]/n:.0f} adjoint {100.0*s['num_grad_evals'][0]/n:.0f}] "
          f"top {s['delta_y_norm_over_c'][0]/n:.0f} post {s['f2_norm'][0]/n:.0f}")
