"""Where a pass of the two-stage kernel goes (s_memtime ticks per section, fenced by s_waitcnt: upper bounds that do not overlap as they do
in the product).  Needs the instrumented build:  make -C mpc_trajectory_generator_amd/csrc -B libnmpc_tk.so OUT=libnmpc_tk.so EXTRA=-DNMPC2_TICKS
then  NMPC_TEAM_HELP=0 NMPC_LIB_PATH=mpc_trajectory_generator_amd/csrc/libnmpc_tk.so python scripts/hyb2_sections.py"""
import sys, numpy as np
sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes
cfg = named_config("cfg2")
P = synthetic_batch(cfg, 11, 64, 0, routes=random_routes(cfg, 11, 32, seed=1000))
sol = BatchSolver(cfg, max_batch=64)
st = sol.solve(P)[2]
b = int(np.argmax(st["reserved"]))
s = sol.solve(P[b:b+1])[2]
n = int(s["reserved"][0])
names = ["top(state machine, two-loop)", "transfer to eval layout", "eval_psi2", "grad store+sync+psi", "consume"]
vals = [s["last_problem_norm_fpr"][0], s["delta_y_norm_over_c"][0], s["f2_norm"][0], s["penalty"][0], s["cost"][0]]
tot = sum(vals)
print("passes", n, "ms", sol.last_batch_ms, "ticks/pass total", tot / n)
for nm, v in zip(names, vals): print(f"  {nm:32s} {v/n:9.1f} ticks/pass  {100*v/tot:5.1f} %")
