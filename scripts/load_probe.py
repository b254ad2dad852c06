"""Microseconds per pass of ONE hard instance as a function of how many copies of it run at the same time (1 ... all wave slots): what the
waves of a CU / the chip cost each other.  Helpers off, scheduling off.  usage: python scripts/load_probe.py cfgN"""
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
os.environ["NMPC_TEAM_HELP"] = "0"; os.environ["NMPC_SCHED"] = "0"
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
name = sys.argv[1]
cfg = named_config(name)
P = synthetic_batch(cfg, 11, 256, 0, routes=random_routes(cfg, 11, 32, seed=1000), synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
s = BatchSolver(cfg, max_batch=4096)
st = s.solve(P)[2]
b = int(np.argmax(st["reserved"]))
out = {"cfg": name, "kernel": s.kernel_name, "passes": int(st["reserved"][b]), "us_per_pass": {}}
for n, owners in ((1, "1"), (256, "1"), (512, "2"), (1024, "4"), (2048, "4")):
    os.environ["NMPC_TEAM_OWNERS"] = owners
    s2 = BatchSolver(cfg, max_batch=4096)
    Pn = np.repeat(P[b:b + 1], n, axis=0)
    s2.solve(Pn); st2 = s2.solve(Pn)[2]
    out["us_per_pass"][f"{n} copies ({owners} per CU x {min(n, 256 * int(owners)) // int(owners)} CUs)"] = round(1e3 * s2.last_batch_ms / int(st2["reserved"][0]) / max(1, n // (256 * 4 * (2 if name != 'cfg2' else 1))) , 3)
    s2.close()
print(json.dumps(out))
