#!/usr/bin/env python3
"""BASELINE config 4: smooth_velocity weights, per-robot randomised dynamic ellipses, B robots advanced
for K receding-horizon steps (num_steps_taken = 2), warm starts carried, p rebuilt every step.
Default: the whole loop on device (nmpc_loop_*: assembly, solve and state advance are kernels, nothing
crosses PCIe between steps).  --host: parameter assembly on the host (NumPy-vectorised) around
BatchSolver.solve.  Prints one JSON line."""
import argparse
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mpc_trajectory_generator_amd import named_config                            # noqa: E402
from mpc_trajectory_generator_amd import harness                                 # noqa: E402
from mpc_trajectory_generator_amd.solver import BatchSolver                      # noqa: E402
from mpc_trajectory_generator_amd.trajectory import VectorizedRecedingHorizon    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--scene", type=int, default=11)
ap.add_argument("--host", action="store_true")
ap.add_argument("--split", type=int, default=2,
                help="device loop only: split the fleet into this many sub-fleets, each with its own handle and HIP "
                     "stream, stepped alternately -- one sub-fleet's slow instances overlap the others' bulk")
ap.add_argument("--budget", type=int, default=0,
                help="max_total_inner per solve: the deterministic counterpart of the reference's 0.5 s max_duration "
                     "(src/mpc/mpc_generator.py:9,186); 0 = off")
ap.add_argument("--experiments", action="store_true", help="the experiments build of the library (reads the NMPC_* knobs: A/B runs only)")
args = ap.parse_args()
sopts = {"max_total_inner": args.budget} if args.budget > 0 else {}
if args.experiments:
    sopts["experiments"] = True
cfg = named_config("cfg4")
route = harness.scene_route(cfg, args.scene)
rng = np.random.Generator(np.random.PCG64(0))
B, n = args.batch, len(route.x_ref)
i0 = rng.integers(0, max(1, n - 60), B)
starts = np.stack([np.array(route.x_ref)[i0] + rng.normal(0, 0.05, B), np.array(route.y_ref)[i0] + rng.normal(0, 0.05, B),
                   np.array(route.theta_ref)[i0] + rng.normal(0, 0.1, B)], axis=1)
K = cfg.Ndynobs
jj = np.minimum(n - 1, i0[:, None] + rng.integers(0, 30, (B, K)))
c = np.stack([np.array(route.x_ref)[jj], np.array(route.y_ref)[jj]], axis=2)
dyn = (c + rng.uniform(-5, 5, (B, K, 2)), c + rng.uniform(-5, 5, (B, K, 2)), rng.uniform(0.05, 0.1, (B, K)),
       rng.uniform(0.3, 1.0, (B, K)), rng.uniform(0.3, 1.0, (B, K)), rng.uniform(0, np.pi, (B, K)))
solver = BatchSolver(cfg, max_batch=B, **sopts)
if not args.host:
    from mpc_trajectory_generator_amd.trajectory import DeviceRecedingHorizon
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    parts = np.array_split(np.arange(B), args.split)
    loops, streams = [], []
    for ids in parts:
        sv = solver if not loops else BatchSolver(cfg, max_batch=len(ids), **sopts)
        loops.append(DeviceRecedingHorizon(sv, route, starts[ids], tuple(a[ids] for a in dyn), max_steps=args.steps, idx0=i0[ids]))
        strm = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(strm)) == 0
        streams.append(strm)
    for rh, strm in zip(loops, streams):
        rh.step(strm)                           # step 0 = cold start; timed separately
    st0 = np.concatenate([rh.read()[4] for rh in loops])        # (synchronises)
    t0 = time.perf_counter()
    for k in range(1, args.steps):
        for rh, strm in zip(loops, streams):
            rh.step(strm)
    outs = [rh.read() for rh in loops]          # synchronises
    total = time.perf_counter() - t0
    done = np.concatenate([o[3] for o in outs])
    st = np.concatenate([o[4] for o in outs])
    from mpc_trajectory_generator_amd import _lib
    if hasattr(_lib.load_library(), "nmpc_debug_win_stats"):        # instrumented build (-DNMPC_WIN_STATS, scripts/win_stats.py)
        buf = (ctypes.c_ulonglong * 2)()
        _lib.load_library().nmpc_debug_win_stats(buf, 0)
        print(f"windowed cross-track searches {buf[0]}, fell back {buf[1]} ({100.0 * buf[1] / max(buf[0], 1):.1f} %)", file=sys.stderr)
    print(json.dumps({
        "metric": "nmpc_receding_horizon_solves_per_sec", "value": B * (args.steps - 1) / total, "unit": "solves/s",
        "config": {"workload": f"cfg4 smooth_velocity, scene {args.scene}, B={B}, {args.steps} receding-horizon steps, "
                               "num_steps_taken=2, warm start (u, y carried; c reset), loop entirely on device"
                               + (f", at most {args.budget} PANOC iterations per solve (NotConvergedOutOfTime beyond)" if args.budget else "")
                               + (f", fleet split into {args.split} sub-fleets on {args.split} streams" if args.split > 1 else ""),
                   "kernel": solver.kernel_name},
        "ms_per_step": 1e3 * total / (args.steps - 1), "mean_inner_iters_first_step": float(st0["num_inner_iterations"].mean()),
        "mean_inner_iters_last_step": float(st["num_inner_iterations"].mean()),
        "converged_frac_last_step": float((st["exit_status"] == 0).mean()),
        "out_of_time_frac_last_step": float((st["exit_status"] == 2).mean()), "robots_at_goal": int(done.sum())}))
    sys.exit(0)
rh = VectorizedRecedingHorizon(route, starts, dyn)
rh.idx = i0.astype(np.int64)
t_solve, t_asm, iters, conv = [], [], [], []


def solve(P, U, Y):
    t = time.perf_counter()
    out = solver.solve(P, u0=U, y0=Y)
    t_solve.append(time.perf_counter() - t)
    return out


t0 = time.perf_counter()
for k in range(args.steps):
    t = time.perf_counter()
    P = rh.assemble()
    t_asm.append(time.perf_counter() - t)
    U, Y, st = solve(P, rh.U, rh.Y)
    rh.U, rh.Y = U, Y
    rh.advance(U)
    iters.append(float(st["num_inner_iterations"].mean()))
    conv.append(float((st["exit_status"] == 0).mean()))
total = time.perf_counter() - t0
print(json.dumps({
    "metric": "nmpc_receding_horizon_solves_per_sec", "value": B * args.steps / total, "unit": "solves/s",
    "config": {"workload": f"cfg4 smooth_velocity, scene {args.scene}, B={B}, {args.steps} receding-horizon steps, "
                           "num_steps_taken=2, warm start (u, y carried; c reset), host-side vectorised p assembly"},
    "solver_only_solves_per_sec": B * args.steps / sum(t_solve), "ms_per_step_total": 1e3 * total / args.steps,
    "ms_per_step_solve_incl_pcie": 1e3 * float(np.mean(t_solve)), "ms_per_step_assembly": 1e3 * float(np.mean(t_asm)),
    "mean_inner_iters_first_step": iters[0], "mean_inner_iters_steady": float(np.mean(iters[5:])),
    "converged_frac_steady": float(np.mean(conv[5:])), "robots_at_goal": int(rh.done.sum())}))
