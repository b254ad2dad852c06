"""Repeat full-batch and small-batch solves many times and demand identical bits every time: the team protocol (LDS
mailboxes, claim / done flags) and the migration pool are timing-dependent in WHERE work runs, never in what comes out.
usage: python scripts/stress_teams.py [reps]"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
out = {}
for name, B in (("cfg1", 8192), ("cfg1", 700), ("cfg1", 96), ("cfg2", 2048), ("cfg2", 40), ("cfg3", 3000), ("cfg4", 512)):
    cfg = named_config(name)
    P = synthetic_batch(cfg, 11, B, 3, routes=random_routes(cfg, 11, 8, seed=77), synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
    sol = BatchSolver(cfg, max_batch=B)
    digests = set()
    for r in range(reps if B < 4000 else max(3, reps // 4)):
        u, y, st = sol.solve(P)
        h = hashlib.sha256()
        h.update(u.tobytes()); h.update(y.tobytes())
        for f in ("exit_status", "num_inner_iterations", "num_cost_evals", "num_grad_evals", "reserved", "cost", "penalty", "f2_norm"):
            h.update(np.ascontiguousarray(st[f]).tobytes())
        digests.add(h.hexdigest()[:16])
    out[f"{name}_B{B}"] = sorted(digests)
    sol.close()
print(json.dumps(out))
assert all(len(v) == 1 for v in out.values()), "non-deterministic results"
print("STRESS_OK")
