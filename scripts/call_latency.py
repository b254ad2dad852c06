"""End-to-end latency of one B = 1 solve through the reference-shaped handle (tcp_shim.OptimizerTcpManager.call: Python list in,
Python list out) against the kernel time inside it -- the reference's own call pattern (src/mpc/mpc_generator.py:206).
usage: python scripts/call_latency.py"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.tcp_shim import OptimizerTcpManager
from mpc_trajectory_generator_amd.harness import synthetic_batch
cfg = named_config("cfg1")
P = synthetic_batch(cfg, 11, 64, 0)
mng = OptimizerTcpManager(config=cfg)
mng.start(); mng.ping()
out = []
for b in range(16):
    p = P[b].tolist()
    mng.call(p)                                # warm start state differs per call; time cold-ish calls with a reset in between
    t0 = time.perf_counter(); r = mng.call(p, initial_guess=[0.0] * cfg.n_u); t1 = time.perf_counter()
    g = r.get()
    out.append((1e3 * (t1 - t0), g.solve_time_ms, g.num_inner_iterations))
mng.kill()
wall = np.array([o[0] for o in out]); kern = np.array([o[1] for o in out])
print(json.dumps({"calls": len(out), "wall_ms_median": round(float(np.median(wall)), 3), "kernel_ms_median": round(float(np.median(kern)), 3),
                  "overhead_ms_median": round(float(np.median(wall - kern)), 3), "overhead_ms_max": round(float((wall - kern).max()), 3),
                  "iters_median": int(np.median([o[2] for o in out]))}))
