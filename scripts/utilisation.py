"""How much of a batch's wave-slot time is idle, and is the batch bounded by its longest instance?  Start / finish of every instance on the
100 MHz clock (NMPC_DEBUG_PRIO: 2 for the one-stage kernel, 1 for the two-stage one), with the step-aside scheduling OFF: the diagnosis that led to it
(round 4: 71 / 82 / 77 / 82 % busy for configs 1-4; bounds max(work, longest instance) = 42 / 141 / 85 / 94 ms against 48 / 173 / 99 / 114 measured).  usage: python scripts/utilisation.py cfgN [seed]"""
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
name = sys.argv[1]; seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
os.environ["NMPC_DEBUG_PRIO"] = "1" if name == "cfg2" else "2"
os.environ.setdefault("NMPC_SCHED", "0")      # an instance's start -> finish time is busy time only while instances keep their waves
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
cfg = named_config(name); B = 8192
P = synthetic_batch(cfg, 11, B, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed), synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
s = BatchSolver(cfg, max_batch=B, experiments=True)
s.solve(P); u, y, st = s.solve(P)
t0 = st["delta_y_norm_over_c"].min()
start = (st["delta_y_norm_over_c"] - t0) * 1e-5; end = (st["cost"] - t0) * 1e-5
S = 1024 if name == "cfg2" else 2048
span = float(end.max()); busy = float((end - start).sum())
dur = end - start; ps = st["reserved"].astype(np.int64)
k = int(np.argmax(dur))
print(json.dumps({"cfg": name, "kernel_ms": s.last_batch_ms, "span_ms": span, "slots": S, "utilisation": busy / (S * span), "work_bound_ms": busy / S,
                  "longest_instance_ms": float(dur.max()), "longest_instance_passes": int(ps[k]), "longest_instance_start_ms": float(start[k]),
                  "queue_dry_ms": float(start.max()), "us_per_pass_mean": float(1e3 * busy / ps.sum()),
                  "us_per_pass_long(>6000)": float(1e3 * dur[ps > 6000].sum() / ps[ps > 6000].sum()),
                  "us_per_pass_short(<1500)": float(1e3 * dur[ps < 1500].sum() / ps[ps < 1500].sum())}))
