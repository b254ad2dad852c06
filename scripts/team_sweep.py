"""Kernel ms of a batch (seeds 0, 1, 2) under the team knobs given as KEY=VAL,... sets on the command line.
usage: python scripts/team_sweep.py cfg1 "NMPC_TEAM_RECRUIT=1,NMPC_TEAM_WANT=3000" "NMPC_TEAM_RECRUIT=2,NMPC_TEAM_WANT=4000" ..."""
import json, os, subprocess, sys
sys.path.insert(0, ".")
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import numpy as np
    from mpc_trajectory_generator_amd import named_config
    from mpc_trajectory_generator_amd.solver import BatchSolver
    from mpc_trajectory_generator_amd.harness import synthetic_batch
    from mpc_trajectory_generator_amd.frontend import random_routes
    name = sys.argv[1]
    B = int(os.environ.get("SWEEP_B", "8192"))
    cfg = named_config(name)
    sol = BatchSolver(cfg, max_batch=B)
    out = {}
    for seed in (0, 1, 2):
        P = synthetic_batch(cfg, 11, B, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed),
                            synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
        sol.solve(P)
        ms = []
        for _ in range(3):
            st = sol.solve(P)[2]
            ms.append(sol.last_batch_ms)
        out[f"s{seed}"] = round(min(ms), 2)
        out[f"s{seed}_mean"] = round(float(np.mean(ms)), 2)
        if seed == 0:
            out["checksum"] = float(st["num_inner_iterations"].astype(np.float64).sum() + st["cost"].sum())
            out["slowest_ms"] = round(float(st["solve_time_ms"].max()), 2)
    out["mean"] = round((out["s0"] + out["s1"] + out["s2"]) / 3, 2)
    print(json.dumps(out))
    sys.exit(0)
name = sys.argv[1]
for spec in sys.argv[2:]:
    env = dict(kv.split("=") for kv in spec.split(",") if kv)
    r = subprocess.run([sys.executable, __file__, name, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
    print(json.dumps({"env": env, "res": json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-400:]}), flush=True)
