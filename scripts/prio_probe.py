"""Does s_setprio let one wave of a SIMD run at lone speed?  2048 copies of one hard instance (every wave slot
busy with identical work); NMPC_DEBUG_PRIO=1 gives the waves in even hardware slots priority 3 and those in odd slots 0, =2 leaves
all at 0.  Prints the per-wave cycle counts by slot parity."""
import os
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
for B in (1, 1024, 2048):
    Pb = np.repeat(P[4175:4176], B, axis=0)
    sol.solve(Pb)
    _, _, s = sol.solve(Pb)
    c = s["last_problem_norm_fpr"]
    slot = s["f2_norm"].astype(int)                      # hardware wave slot within the SIMD (debug fields of the status)
    ev, od = c[slot % 2 == 0], c[slot % 2 == 1]
    print("  wave slots used:", np.bincount(slot))
    print(f"dbg={os.environ.get('NMPC_DEBUG_PRIO')} B={B}: kernel {sol.last_batch_ms:.1f} ms, {s['reserved'][0]} passes, "
          f"{1e3 * sol.last_batch_ms / s['reserved'][0]:.2f} us/pass; cycles/1e6 even: mean {ev.mean()/1e6:.1f} min {ev.min()/1e6:.1f} max {ev.max()/1e6:.1f}"
          + (f" | odd: mean {od.mean()/1e6:.1f} min {od.min()/1e6:.1f} max {od.max()/1e6:.1f}" if len(od) else ""))
