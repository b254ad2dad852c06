"""Same instance replicated B times: kernel time vs number of concurrent waves (CU-level contention)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
cfg = named_config("cfg1")
sol = BatchSolver(cfg, max_batch=8192)
P = synthetic_batch(cfg, 11, 512, 0)
p = P[460:461]
for B in (1, 64, 256, 512, 1024, 2048, 4096, 8192):
    Pb = np.repeat(p, B, axis=0)
    sol.solve(Pb)
    _, _, s = sol.solve(Pb)
    print(f"B={B:5d}: kernel {sol.last_batch_ms:8.2f} ms  passes/inst {s['reserved'][0]}  -> {1e3*sol.last_batch_ms/s['reserved'][0]/max(1,-(-B//2048)):.2f} us/pass/round")
