import sys
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
cfg = named_config("cfg1")
sol = BatchSolver(cfg, max_batch=8192)
P = synthetic_batch(cfg, 11, 512, 0)
for b in (460, 337, 423):
    sol.solve(P[b:b + 1])
    _, _, s = sol.solve(P[b:b + 1])
    n = s["reserved"][0]
    print(f"inst {b}: passes {n} iters {s['num_inner_iterations'][0]} kernel {s['solve_time_ms'][0]:.2f} ms | per pass cycles: "
          f"eval {s['last_problem_norm_fpr'][0]/n:.0f} top(begin_step etc) {s['delta_y_norm_over_c'][0]/n:.0f} post(state) {s['f2_norm'][0]/n:.0f}")
