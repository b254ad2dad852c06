"""us per pass of one hard instance solved alone without helpers (a single wave on the chip) and kernel ms of a full batch,
for the library in NMPC_LIB_PATH.  usage: python scripts/lone_probe.py cfg2 [B]"""
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
cfg = named_config(name)
P = synthetic_batch(cfg, 11, B, 0, routes=random_routes(cfg, 11, 32, seed=1000), synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
sol = BatchSolver(cfg, max_batch=B)
st = sol.solve(P)[2]
ms = min((sol.solve(P), sol.last_batch_ms)[1] for _ in range(2))
b = int(np.argmax(st["reserved"]))
sol.solve(P[b:b + 1])
s1 = sol.solve(P[b:b + 1])[2]
print(json.dumps({"lib": os.environ.get("NMPC_LIB_PATH", "default"), "help": os.environ.get("NMPC_TEAM_HELP", "1"), "batch_ms": round(ms, 2),
                  "lone_ms": round(sol.last_batch_ms, 2), "passes": int(s1["reserved"][0]), "us_per_pass": round(1e3 * sol.last_batch_ms / int(s1["reserved"][0]), 3),
                  "checksum": float(st["num_inner_iterations"].astype(np.float64).sum() + st["cost"].sum())}))
