"""Full batches of every BASELINE config, several seeds: two launches (the second on a permuted batch) must give the same bits, and a sample must equal the oracle's.
The long version of tests/test_gpu_fullbatch.py's permutation test -- what catches a scheduler-dependent miscompilation (csrc/Makefile).  usage: python scripts/determinism_check.py [seeds...]"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import oracle_for, STATUS_FIELDS
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
seeds = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
bad = 0
for name in ("cfg1", "cfg2", "cfg3", "cfg4"):
    cfg = named_config(name)
    sol = BatchSolver(cfg, max_batch=8192)
    for seed in seeds:
        P = synthetic_batch(cfg, 11, 8192, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed), synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
        u, y, st = sol.solve(P)
        perm = np.random.default_rng(seed).permutation(8192)
        u2, y2, st2 = sol.solve(P[perm])
        same = np.array_equal(u2, u[perm]) and np.array_equal(y2, y[perm]) and all(np.array_equal(st2[f], st[f][perm]) for f in STATUS_FIELDS)
        idx = np.random.default_rng(100 + seed).choice(8192, 48, replace=False)
        uo, yo, sto = oracle_for(cfg).solve_batch(P[idx], threads=16)
        par = np.array_equal(u[idx], uo) and np.array_equal(y[idx], yo) and all(np.array_equal(st[f][idx], sto[f]) for f in STATUS_FIELDS)
        print(name, "seed", seed, sol.kernel_name, "permutation-invariant", same, "sample == oracle", par, "ms", round(sol.last_batch_ms, 1), flush=True)
        bad += (not same) + (not par)
    sol.close()
print("DETERMINISM_OK" if bad == 0 else f"FAILURES {bad}")
sys.exit(1 if bad else 0)
