"""Headline batch with NMPC_DEBUG_PRIO=2: which hardware wave slot each instance ran on, how fast, and when it
finished (the status fields last_problem_norm_fpr / f2_norm / cost carry cycles / slot / 100 MHz finish time)."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
name = sys.argv[2] if len(sys.argv) > 2 else "cfg1"
cfg = named_config(name)
sol = BatchSolver(cfg, max_batch=8192)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
P = synthetic_batch(cfg, 11, 8192, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed), synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
sol.solve(P)
u, y, st = sol.solve(P)
ps = st["reserved"].astype(np.int64); cyc = st["last_problem_norm_fpr"]; slot = st["f2_norm"].astype(int)
print("kernel ms", sol.last_batch_ms, "slots", np.bincount(slot))
t0 = st["delta_y_norm_over_c"].min()
end = (st["cost"] - t0) / 100e3          # ms since the first instance started (100 MHz clock)
first = (st["delta_y_norm_over_c"] - t0) / 100e3
moves = st["penalty"].astype(int)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(f"gpurun_out/slots_{name}_{seed}.npz", passes=ps, first=first, end=end, slot=slot, moves=moves, cyc=cyc, iters=st["num_inner_iterations"], outer=st["num_outer_iterations"])
last = np.argsort(-end)[:10]
print("  last to finish:", [(int(b), int(ps[b]), int(slot[b]), round(float(first[b]), 1), int(moves[b]), round(float(end[b]), 1)) for b in last], "(inst, passes, slot, first start ms, migrations, end ms)")
print("  start time of instances > 6000 passes: pctl", np.percentile(first[ps > 6000], [0, 50, 90, 100]).round(1), " migrations hist", np.bincount(moves[ps > 6000]))
top = np.argsort(-ps)[:12]
for b in top:
    print(f"  inst {b}: passes {ps[b]} final slot {slot[b]}, first started {first[b]:.1f} ms, migrations {moves[b]}, last leg {cyc[b]/2.4e6:.1f} ms, finished at {end[b]:.1f} ms -> {1e3*(end[b]-first[b])/ps[b]:.2f} us/pass overall")
for s_ in (0, 1):
    m = (slot == s_) & (ps > 3000)
    print(f"  slot {s_}: instances>3000 passes: {m.sum()}, mean us/pass {np.mean(cyc[m]/2.4e3/ps[m]):.2f}")
