"""Kernel ms and checksum of a batch of an arbitrary shape for one or more library builds (A/B of kernels outside the BASELINE configs).
usage: python scripts/shape_probe.py N_hor Nobs B lib1.so [lib2.so ...]"""
import json, os, subprocess, sys
sys.path.insert(0, ".")
if sys.argv[1] == "child":
    import numpy as np
    from mpc_trajectory_generator_amd.config import load_config
    from mpc_trajectory_generator_amd.solver import BatchSolver
    from mpc_trajectory_generator_amd.harness import synthetic_batch
    from mpc_trajectory_generator_amd.frontend import random_routes
    N, nobs, B = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    cfg = load_config(N_hor=N, Nobs=nobs)
    P = synthetic_batch(cfg, 11, B, 0, routes=random_routes(cfg, 11, 32, seed=1000))
    sol = BatchSolver(cfg, max_batch=B)
    sol.solve(P)
    ms = []
    for _ in range(3):
        st = sol.solve(P)[2]
        ms.append(sol.last_batch_ms)
    print(json.dumps({"kernel": sol.kernel_name, "ms": round(min(ms), 2), "mean_iters": float(st["num_inner_iterations"].mean()),
                      "checksum": float(st["num_inner_iterations"].astype(np.float64).sum() + st["cost"].sum())}))
    sys.exit(0)
for lib in sys.argv[4:]:
    r = subprocess.run([sys.executable, __file__, "child", *sys.argv[1:4]], env=dict(os.environ, NMPC_LIB_PATH=os.path.abspath(lib)), capture_output=True, text=True)
    print(lib, r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-600:], flush=True)
