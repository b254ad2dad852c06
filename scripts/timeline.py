"""Timeline of a helped iteration (one instance alone in its workgroup, three helper waves): cycles from the start of a PANOC step to
fifteen events on the owner's and the first helper's side, median over 63 consecutive steps.  Needs a library built with -DNMPC_TL:
    make -C mpc_trajectory_generator_amd/csrc -B OUT=variants/libnmpc_tl.so EXTRA="-DNMPC_TL -DNMPC_EXPERIMENTS"
    NMPC_LIB_PATH=mpc_trajectory_generator_amd/csrc/variants/libnmpc_tl.so python scripts/timeline.py [instance]
(every event costs the wave ~100 cycles: the sum is above the plain build's iteration)."""
import ctypes, json, os, sys
sys.path.insert(0, ".")
import numpy as np
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
cfg = named_config("cfg1")
P = synthetic_batch(cfg, 11, 8192, 0, routes=random_routes(cfg, 11, 32, seed=1000))
sol = BatchSolver(cfg, max_batch=8192)
lib = ctypes.CDLL(os.environ["NMPC_LIB_PATH"])
buf = (ctypes.c_longlong * (64 * 16))()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 170
sol.solve(P[b:b + 1]); sol.solve(P[b:b + 1])
assert lib.nmpc_debug_timeline(buf) == 0
T = np.array(buf[:], dtype=np.int64).reshape(64, 16)
names = ["begin", "batch_done", "update_done", "direction_done", "post_start", "posted", "eval_start", "eval_done", "own_trials_done", "helper_seen",
         "consumed", "helper_claimed", "helper_eval_start", "helper_eval_done", "helper_flag"]
rows = []
for i in range(63):
    t = T[i]
    if t[0] == 0 or t[10] == 0 or t[11] == 0 or T[i + 1][0] == 0: continue      # (a step without a helped line search)
    r = {n: int(t[j] - t[0]) for j, n in enumerate(names) if t[j] != 0}
    r["next_begin"] = int(T[i + 1][0] - t[0])
    rows.append(r)
print(json.dumps({"instance": b, "kernel_ms": round(sol.last_batch_ms, 3), "steps": len(rows),
                  "median_cycles": {k: int(np.median([r[k] for r in rows if k in r])) for k in names + ["next_begin"] if any(k in r for r in rows)}}))
