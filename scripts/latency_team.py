"""Small-batch latency of the hybrid kernel's teams: kernel ms of B = 1 / 24 / 512 instances
with helpers (default) and without (NMPC_TEAM_HELP=0), and the CPU oracle's single-thread time for the same instance.
usage: python scripts/latency_team.py [cfg]"""
import json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, ".")
if len(sys.argv) > 2 and sys.argv[2] == "child":
    from mpc_trajectory_generator_amd import named_config
    from mpc_trajectory_generator_amd.solver import BatchSolver
    from mpc_trajectory_generator_amd.harness import synthetic_batch
    from mpc_trajectory_generator_amd.frontend import random_routes
    name = sys.argv[1]
    cfg = named_config(name)
    P = synthetic_batch(cfg, 11, 512, 0, routes=random_routes(cfg, 11, 32, seed=1000),
                        synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
    sol = BatchSolver(cfg, max_batch=512, experiments=True)
    st = sol.solve(P)[2]
    out = {"B512_ms": round(min((sol.solve(P), sol.last_batch_ms)[1] for _ in range(3)), 3)}
    order = np.argsort(-st["reserved"].astype(np.int64))
    hard, mid = int(order[0]), int(order[len(order) // 2])
    for tag, b in (("hardest", hard), ("median", mid)):
        ms = min((sol.solve(P[b:b + 1]), sol.last_batch_ms)[1] for _ in range(3))
        out[f"B1_{tag}_ms"] = round(ms, 3)
        out[f"B1_{tag}_iters"] = int(st["num_inner_iterations"][b])
        out[f"B1_{tag}_passes3"] = int(st["reserved"][b])
    out["B24_ms"] = round(min((sol.solve(P[:24]), sol.last_batch_ms)[1] for _ in range(3)), 3)
    out["ids"] = [hard, mid]
    print(json.dumps(out))
    sys.exit(0)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
res = {}
for tag, env in (("team", {}), ("no_help", {"NMPC_TEAM_HELP": "0"})):
    r = subprocess.run([sys.executable, __file__, name, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
    res[tag] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-500:]
try:   # the CPU port on one thread, the same two instances (oracle = test infrastructure; this is a probe, not the product)
    from oracle import Oracle
    from mpc_trajectory_generator_amd import named_config
    from mpc_trajectory_generator_amd.harness import synthetic_batch
    from mpc_trajectory_generator_amd.frontend import random_routes
    cfg = named_config(name)
    P = synthetic_batch(cfg, 11, 512, 0, routes=random_routes(cfg, 11, 32, seed=1000),
                        synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
    o = Oracle(cfg.N_hor, cfg.Nobs, cfg.Ndynobs, cfg.ts, cfg.lin_vel_min, cfg.lin_vel_max, cfg.ang_vel_max,
               cfg.lin_acc_min, cfg.lin_acc_max, cfg.ang_acc_max)
    for tag, b in zip(("hardest", "median"), res["team"]["ids"]):
        t = time.perf_counter(); o.solve_batch(P[b:b + 1], threads=1); res[f"cpu_1thread_{tag}_ms"] = round(1e3 * (time.perf_counter() - t), 2)
except Exception as e:      # noqa: BLE001
    res["cpu"] = repr(e)
res["config"] = name
print(json.dumps(res))
