"""Where does a build of nmpc_solve_hyb2_kernel first leave the oracle?  Config-2 instances under tiny iteration caps, HIP (the library in
NMPC_LIB_PATH) against the oracle with the same options, field by field.  usage: NMPC_LIB_PATH=... python scripts/hyb2_caps.py tag [n]"""
import json, os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import oracle_for, STATUS_FIELDS
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes

tag = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = named_config("cfg2")
P = synthetic_batch(cfg, 11, n, 0, routes=random_routes(cfg, 11, 32, seed=1000))
for help_ in ("0", "1"):
    os.environ["NMPC_TEAM_HELP"] = help_
    for (mo, mi) in ((1, 1), (1, 2), (1, 3), (1, 4), (1, 6), (1, 12), (1, 30), (2, 4), (2, 30), (3, 100), (10, 500)):
        s = BatchSolver(cfg, max_batch=n, max_outer=mo, max_inner=mi)
        u, y, st = s.solve(P)
        uo, yo, sto = oracle_for(cfg, **s.oracle_opts()).solve_batch(P, threads=16)
        s.close()
        bad_u = int(np.any(u != uo, axis=1).sum()); bad_y = int(np.any(y != yo, axis=1).sum())
        bad_f = {f: int((st[f] != sto[f]).sum()) for f in STATUS_FIELDS if (st[f] != sto[f]).any()}
        err_u = float(np.abs(u - uo).max())
        print(json.dumps({"lib": tag, "help": help_, "max_outer": mo, "max_inner": mi, "n": n, "bad_u": bad_u, "bad_y": bad_y, "max_abs_du": err_u, "bad_fields": bad_f}), flush=True)
