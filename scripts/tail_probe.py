import sys
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
cfg = named_config("cfg1")
sol = BatchSolver(cfg, max_batch=8192)
from mpc_trajectory_generator_amd.frontend import random_routes
P = synthetic_batch(cfg, 11, 8192, 0, routes=random_routes(cfg, 11, 32, seed=1000))
u, y, st = sol.solve(P)
ps = st["reserved"].astype(np.int64)
print("passes: mean", ps.mean(), "p50", np.median(ps), "p90", np.percentile(ps, 90), "p99", np.percentile(ps, 99), "max", ps.max(), "sum", ps.sum())
print("evals/iter mean", ((st["num_cost_evals"] + st["num_grad_evals"]).sum() / st["num_inner_iterations"].sum()))
order = np.argsort(-ps)[:6]
for b in order:
    _, _, s = sol.solve(P[b:b + 1])
    print(f"inst {b}: passes {ps[b]} iters {st['num_inner_iterations'][b]} outer {st['num_outer_iterations'][b]} grad evals {st['num_grad_evals'][b]} cost evals {st['num_cost_evals'][b]} "
          f"penalty {st['penalty'][b]:.3g} exit {st['exit_status'][b]} lone {sol.last_batch_ms:.1f} ms")
hist = np.histogram(ps, bins=[0, 500, 1000, 2000, 4000, 6000, 8000, 10000, 12000, 16000, 20000])
print(hist)
