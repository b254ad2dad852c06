"""Where an iteration of the hybrid kernel goes (cycles per iteration by section of its loop), for single instances solved alone with
and without helper waves.  Needs a library built with -DNMPC_PROF2 (the section timers of nmpc_solve_hyb.h):
    make -C mpc_trajectory_generator_amd/csrc -B OUT=variants/libnmpc_prof2.so EXTRA="-DNMPC_PROF2 -DNMPC_EXPERIMENTS"      (=2: the evaluation's own sections)
    NMPC_LIB_PATH=mpc_trajectory_generator_amd/csrc/variants/libnmpc_prof2.so python scripts/sections.py [cfgN] [ids...]
Sections: 0 phase handlers in front of the batch, 1 the batch of inner products, 2 exit test / L-BFGS update, 3 recurrences + direction,
4 envelope / trial points / request, 5 evaluation, 6 consumption of the trials."""
import json
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
ids = [int(x) for x in sys.argv[2:]] or [170, 330]
cfg = named_config(name)
kw = dict(synthetic_circles=(name == "cfg3"), random_dyn=(name == "cfg4"))
P = synthetic_batch(cfg, 11, 8192, 0, routes=random_routes(cfg, 11, 32, seed=1000), **kw)
sol = BatchSolver(cfg, max_batch=8192)
for b in ids:
    sol.solve(P[b:b + 1])
    st = sol.solve(P[b:b + 1])[2]
    it = float(st["num_inner_iterations"][0])
    sec = [float(st[f][0]) for f in ("last_problem_norm_fpr", "delta_y_norm_over_c", "f2_norm", "penalty", "cost", "solve_time_ms")] + [64.0 * float(st["num_cost_evals"][0])]
    print(json.dumps({"config": name, "instance": b, "team_help": os.environ.get("NMPC_TEAM_HELP", "1"), "iterations": int(it), "passes": int(st["reserved"][0]),
                      "kernel_ms": round(sol.last_batch_ms, 3), "cycles_per_iteration": [round(x / it) for x in sec], "sum": round(sum(sec) / it)}))
