"""Do a kernel's results depend on what the registers / LDS held before it started?  The batch is solved after scrubbing every SIMD's
register file and every CU's LDS (tests/scrub/scrub.hip) with different patterns; a difference means an uninitialised read.
usage: [NMPC_LIB_PATH=...] python scripts/scrub_probe.py tag [cfgN ...]"""
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
for name in (sys.argv[2:] or ["cfg2"]):
    cfg = named_config(name)
    B = 8192
    P = synthetic_batch(cfg, 11, B, 0, routes=random_routes(cfg, 11, 32, seed=1000), synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
    s = BatchSolver(cfg, max_batch=B)
    res = []
    for pat in (0x00000000, 0x7ff80000, 0xdeadbeef, 0x00000000):
        rc = scrub.nmpc_scrub(0, ctypes.c_uint(pat), 4096, 160 * 1024)
        assert rc == 0, rc
        res.append(s.solve(P))
    def nd(a, b):
        bad = np.any(a[0] != b[0], axis=1) | np.any(a[1] != b[1], axis=1)
        for f in STATUS_FIELDS:
            bad |= a[2][f] != b[2][f]
        return int(bad.sum())
    print(json.dumps({"lib": tag, "cfg": name, "kernel": s.kernel_name, "zero_vs_nan": nd(res[0], res[1]), "zero_vs_beef": nd(res[0], res[2]), "zero_vs_zero": nd(res[0], res[3])}), flush=True)
    s.close()
