"""Which register does a wrong build read before writing it?  One instance per wave (no wave takes a second instance, so the only
stale content is what the scrub kernel left), registers poisoned in subsets (tests/scrub/scrub.hip, masked variant): a register whose
poison changes the results is read uninitialised.  usage: NMPC_LIB_PATH=... python scripts/scrub_bisect.py tag"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import STATUS_FIELDS
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes

scrub = ctypes.CDLL(os.path.join("tests", "scrub", "libscrub.so"))
tag = sys.argv[1]
cfg = named_config("cfg2")
B = 1024
P = synthetic_batch(cfg, 11, B, 0, routes=random_routes(cfg, 11, 32, seed=1000))
os.environ["NMPC_TEAM_HELP"] = "0"; os.environ["NMPC_TEAM_OWNERS"] = "4"
s = BatchSolver(cfg, max_batch=B, max_outer=2, max_inner=30)

def run(regs, pb=0x7ff80000):
    m = (ctypes.c_uint * 16)()
    for r in regs:
        m[r // 32] |= 1 << (r % 32)
    assert scrub.nmpc_scrub_masked(0, ctypes.c_uint(0), ctypes.c_uint(pb), m, 4096, 160 * 1024) == 0
    return s.solve(P)

def nd(a, b):
    bad = np.any(a[0] != b[0], axis=1) | np.any(a[1] != b[1], axis=1)
    for f in STATUS_FIELDS:
        bad |= a[2][f] != b[2][f]
    return int(bad.sum())

base = run([])
print(json.dumps({"lib": tag, "baseline_repeat_differs": nd(base, run([])), "all_poisoned_differs": nd(base, run(range(512)))}), flush=True)
culprits = []
def search(regs):
    if nd(base, run(regs)) == 0:
        return
    if len(regs) == 1:
        culprits.append(regs[0]); return
    h = len(regs) // 2
    search(regs[:h]); search(regs[h:])
search(list(range(512)))
print(json.dumps({"lib": tag, "culprits": [("v%d" % r if r < 256 else "a%d" % (r - 256)) for r in culprits],
                  "each_differs": [nd(base, run([r])) for r in culprits]}), flush=True)
