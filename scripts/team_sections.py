"""Where does an iteration of a single instance go?  (Needs the instrumentation of profiles/r04/gram_experiment.patch -- the NMPC_PROF2 blocks of
nmpc_solve_hyb.h -- which is not in the shipped sources; GRAM=1 in the environment also needs the patch's lbfgs_gram option.)  Library built with -DNMPC_PROF2 (make -C mpc_trajectory_generator_amd/csrc OUT=... EXTRA=-DNMPC_PROF2,
given in NMPC_LIB_PATH): cycles per real pass at the top of the loop (state machine: two-loop, updates), in the evaluation, after it (trial
bookkeeping, including the time spent waiting for and consuming helpers' results), and how often an iteration needed the helpers.
The status fields are overloaded by that build.  usage: NMPC_LIB_PATH=... python scripts/team_sections.py [cfgN]"""
import json, os, subprocess, sys
import numpy as np
sys.path.insert(0, ".")
if len(sys.argv) > 2 and sys.argv[2] == "child":
    from mpc_trajectory_generator_amd import named_config
    from mpc_trajectory_generator_amd.solver import BatchSolver
    from mpc_trajectory_generator_amd.harness import synthetic_batch
    from mpc_trajectory_generator_amd.frontend import random_routes
    name = sys.argv[1]
    cfg = named_config(name)
    P = synthetic_batch(cfg, 11, 512, 0, routes=random_routes(cfg, 11, 32, seed=1000), synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
    gram = int(os.environ.get("GRAM", "0"))
    sol = BatchSolver(cfg, max_batch=512, lbfgs_gram=gram)
    for b in [int(x) for x in sys.argv[3].split(",")]:
        sol.solve(P[b:b + 1])
        _, _, s = sol.solve(P[b:b + 1])
        real = float(s["cost"][0]); it = int(s["num_inner_iterations"][0])
        print(json.dumps({"inst": b, "help": os.environ.get("NMPC_TEAM_HELP", "1"), "kernel_ms": round(sol.last_batch_ms, 3), "iters": it,
                          "passes_counted": int(s["reserved"][0]), "real_passes": int(real), "deep_iterations": int(s["num_cost_evals"][0]),
                          "deep_served_by_helpers": int(s["num_outer_iterations"][0]), "backtracks": int(s["num_grad_evals"][0]),
                          "cycles_per_real_pass": {"top": round(s["delta_y_norm_over_c"][0] / real), "eval": round(s["last_problem_norm_fpr"][0] / real),
                                                   "post": round(s["f2_norm"][0] / real)},
                          "wait_cycles_per_deep_iteration": round(s["penalty"][0] / max(1, int(s["num_cost_evals"][0]))),
                          "total_cycles": round(float(s["delta_y_norm_over_c"][0] + s["last_problem_norm_fpr"][0] + s["f2_norm"][0])),
                          "gram": gram, "gram_blocks": it if gram else 0,
                          "gram_cycles_per_block": {"batch": round(s["solve_time_ms"][0] / max(1, it)), "chains": round(16.0 * s["exit_status"][0] / max(1, it))} if gram else None}), flush=True)
    sys.exit(0)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
ids = sys.argv[2] if len(sys.argv) > 2 else "170,330"
for env in ({}, {"NMPC_TEAM_HELP": "0"}):
    r = subprocess.run([sys.executable, __file__, name, "child", ids], env=dict(os.environ, **env), capture_output=True, text=True)
    print(r.stdout.strip() if r.returncode == 0 else r.stderr[-800:], flush=True)
