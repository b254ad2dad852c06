"""A/B of the step-aside scheduling (NMPC_SCHED, NMPC_SCHED_THETA) on a config's batch: kernel ms over seeds, checksum of the results (must not move),
slot utilisation.  One subprocess per setting.  usage: python scripts/sched_ab.py cfgN "NMPC_SCHED=0" "NMPC_SCHED=1,NMPC_SCHED_THETA=0.9" ..."""
import json, os, subprocess, sys
CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
name = sys.argv[1]
cfg = named_config(name); B = 8192
s = BatchSolver(cfg, max_batch=B, experiments=True)
out = {"cfg": name, "env": os.environ.get("AB_LABEL"), "ms": [], "checksum": []}
for seed in (0, 1, 2):
    P = synthetic_batch(cfg, 11, B, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed), synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
    s.solve(P)
    best = 1e9
    for _ in range(3):
        u, y, st = s.solve(P); best = min(best, s.last_batch_ms)
    out["ms"].append(round(best, 2)); out["checksum"].append(float(u.sum() + y.sum() + st["num_inner_iterations"].sum()))
print(json.dumps(out))
"""
name = sys.argv[1]
for setting in sys.argv[2:]:
    env = dict(os.environ); env["AB_LABEL"] = setting
    for kv in setting.split(","):
        if kv: k, v = kv.split("="); env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD, name], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-800:], flush=True)
