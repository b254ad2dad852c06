"""One-off stress of the bit-exactness claim: random problem shapes x random solver options / restatement switches x random
batches, HIP (through the C ABI) against the oracle.  usage: python scripts/fuzz_parity.py [n_cases] [seed]"""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import STATUS_FIELDS, oracle_for
from mpc_trajectory_generator_amd.config import load_config
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.solver import BatchSolver

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    N = int(rng.choice([2, 3, 7, 12, 16, 19, 20, 21, 24, 27, 32, 33, 36, 40]))
    nobs = int(rng.choice([0, 1, 4, 10, 13, 33, 50, 64]))
    ndyn = int(rng.integers(0, 4))
    opts = dict(akkt_gradient=int(rng.integers(0, 3)), ls_failure=int(rng.integers(0, 2)), inner_status=int(rng.integers(0, 2)),
                lbfgs_memory=int(rng.choice([1, 3, 10])), max_inner=int(rng.choice([30, 120, 500])), max_outer=int(rng.choice([2, 5, 10])),
                max_total_inner=int(rng.choice([0, 0, 200])))
    cfg = load_config(N_hor=N, Nobs=nobs, Ndynobs=ndyn)
    B = int(rng.integers(1, 14))
    P = synthetic_batch(cfg, 11, B, int(rng.integers(0, 10 ** 6)), random_dyn=ndyn > 0)
    s = BatchSolver(cfg, max_batch=16, **opts)
    try:
        u, y, st = s.solve(P)
        uo, yo, sto = oracle_for(cfg, **s.oracle_opts()).solve_batch(P, threads=8)
        ok = np.array_equal(u, uo) and np.array_equal(y, yo) and all(np.array_equal(st[f], sto[f]) for f in STATUS_FIELDS)
        # warm restart with user penalties
        c0 = rng.choice([1.0, 5.0, 125.0], size=B)
        u2, y2, st2 = s.solve(P, u0=u, y0=y, c0=c0)
        uo2, yo2, sto2 = oracle_for(cfg, **s.oracle_opts()).solve_batch(P, u0=u, y0=y, c0=c0, threads=8)
        ok = ok and np.array_equal(u2, uo2) and np.array_equal(y2, yo2) and all(np.array_equal(st2[f], sto2[f]) for f in STATUS_FIELDS)
    finally:
        s.close()
    bad += not ok
    print(f"case {case}: N={N} nobs={nobs} ndyn={ndyn} B={B} {opts} kernel={'hyb' if N <= 20 else 'two-stage'} -> {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad, "of", n_cases)
sys.exit(1 if bad else 0)
