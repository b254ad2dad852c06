"""Time course of a two-stage-kernel batch (NMPC_DEBUG_PRIO=1: start / finish of every instance on the 100 MHz clock, its wave): when the queue
runs dry, how many waves are busy over time, microseconds per pass by phase, who ends the batch.  usage: python scripts/timeline2.py [cfg2] [seed]"""
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
os.environ["NMPC_DEBUG_PRIO"] = "1"
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = named_config(name)
B = 8192
P = synthetic_batch(cfg, 11, B, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed))
s = BatchSolver(cfg, max_batch=B)
s.solve(P)
u, y, st = s.solve(P)
t0 = st["delta_y_norm_over_c"].min()
start = (st["delta_y_norm_over_c"] - t0) * 1e-5; end = (st["cost"] - t0) * 1e-5      # ms
ps = st["reserved"].astype(np.int64); wave = st["f2_norm"].astype(int)
np.savez_compressed(f"gpurun_out/timeline_{name}_{seed}.npz", start=start, end=end, passes=ps, wave=wave, outer=st["num_outer_iterations"])
grid = np.arange(0, end.max(), 5.0)
busy = [(int(t), int(((start <= t) & (end > t)).sum())) for t in grid]
us = 1e3 * (end - start) / ps
early = start < 20; late = start > 60
print(json.dumps({"kernel_ms": s.last_batch_ms, "span_ms": float(end.max()), "queue_dry_ms": float(start.max()), "waves": int(wave.max() + 1),
                  "busy_instances_over_time": busy, "us_per_pass_all": float(us.mean()), "us_per_pass_started_before_20ms": float(us[early].mean()),
                  "us_per_pass_started_after_60ms": float(us[late].mean()) if late.any() else None,
                  "last_to_finish": [(int(b), int(ps[b]), round(float(start[b]), 1), round(float(end[b]), 1)) for b in np.argsort(-end)[:8]],
                  "long_instances_start_pct_50_90_99_100": [round(float(x), 1) for x in np.percentile(start[ps > 6000], [50, 90, 99, 100])]}))
