"""Single-instance latency probe: kernel time per PANOC iteration / per pass for the slowest instances."""
import sys
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch

cfg = named_config(sys.argv[1] if len(sys.argv) > 1 else "cfg1")
sol = BatchSolver(cfg, max_batch=8192)
P = synthetic_batch(cfg, 11, 512, 0)
u, y, st = sol.solve(P)
it = st["num_inner_iterations"]
order = np.argsort(-it.astype(np.int64))
for b in [order[0], order[1], order[len(order) // 2]]:
    for _ in range(2):
        _, _, s1 = sol.solve(P[b:b + 1])
    ms = sol.last_batch_ms
    evals = int(s1["num_cost_evals"][0]) + int(s1["num_grad_evals"][0])
    print(f"inst {b}: inner {s1['num_inner_iterations'][0]} outer {s1['num_outer_iterations'][0]} evals {evals} "
          f"passes {s1['reserved'][0]} kernel {ms:.2f} ms -> {1e3 * ms / max(1, s1['num_inner_iterations'][0]):.2f} us/iter, "
          f"{1e3 * ms / max(1, s1['reserved'][0]):.2f} us/pass")
for B in (64, 1024, 2048, 4096, 8192):
    Pb = synthetic_batch(cfg, 11, B, 0)
    sol.solve(Pb)
    _, _, s2 = sol.solve(Pb)
    print(f"B={B}: kernel {sol.last_batch_ms:.2f} ms, total iters {s2['num_inner_iterations'].sum()}, max {s2['num_inner_iterations'].max()}")
