#!/usr/bin/env python3
"""bench.py -- NMPC solves/s of the batched HIP solver on BASELINE.json's headline configuration.

One "step" = one cold-start solve (u0 = 0, y0 = 0, c0 = 1) of one batch of B = 8192 independent NMPC
problems per GPU (default.yaml, N = 20, scene 11, 10 circle slots; BASELINE.json configs[1]),
inputs already resident in HBM.  N > 1 GPUs: one process per GPU (torch.distributed, RCCL), each
rank solves its own 8192-instance shard (weak scaling, no data-path collective) and the solutions
are gathered over xGMI inside the step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg1|cfg2|cfg3|cfg4] [--batch B]

Prints ONE JSON line (rank 0).  Besides the driver's contract it carries
  roofline      the bound that actually limits the kernel: f64 vector-ALU issue (78.6 TFLOP/s; the
                f64 MFMA dense peak of MI355X is the same number), algorithmic flops from the
                solver's own evaluation counters / the kernel's HIP-event time
  roofline_hbm  algorithmic HBM bytes per launch / the same time, against 8 TB/s (expected << 1 %)
  cpu_baseline  the CPU oracle (oracle/, kind "port": OpEn itself cannot be built here) timed on the
                host cores on a bounded sample of the same batch, and checked bit-for-bit against
                the GPU result for that sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F64_VALU_TFLOPS = 78.6      # MI355X vector f64 (= f64 MFMA dense peak); MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
# HBM bytes per launch of the cfg1 workload from rocprofv3 PMC passes (profiles/r01/pmc_fetch.csv,
# pmc_write.csv: FETCH_SIZE x 1 KiB x 2 [gfx950 wide-read correction, MI355X_MICROARCH.md] + WRITE_SIZE x 1 KiB)
PMC_TRAFFIC_BYTES = {"cfg1": 2 * 27822.21875 * 1024 + 23941.234375 * 1024}     # FETCH_SIZE (KB) x 2 + WRITE_SIZE (KB)


def flop_model(cfg):
    """Algorithmic flops (fma = 2, everything else incl. sin/cos/div = 1), DESIGN.md section 6."""
    N, Nobs, Ndyn = cfg.N_hor, cfg.Nobs, cfg.Ndynobs
    f_fwd = N * (28 + 7 * Nobs + 15 * Ndyn + 21 * (N - 1)) + 14 * N
    f_bwd = N * (88 + 9 * Nobs + 16 * Ndyn)                                  # adjoint sweep, as implemented
    f_iter = 70 * cfg.n_u                                                    # L-BFGS two-loop + PANOC vector ops
    return f_fwd, f_bwd, f_iter


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg1")
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--scene", type=int, default=11)
    ap.add_argument("--routes", type=int, default=32,
                    help="randomised start/goal pairs planned on the scene (0: the scene's own start -> end route)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the extra two-batches-in-flight measurement (kernel traces then hold the timed steps only)")
    ap.add_argument("--warm", action="store_true",
                    help="warm start: every step starts from the previous step's solution and multipliers "
                         "(SURVEY.md section 8d second timing); the headline metric is the cold start")
    args = ap.parse_args()

    import torch
    from mpc_trajectory_generator_amd import named_config
    from mpc_trajectory_generator_amd.harness import synthetic_batch
    from mpc_trajectory_generator_amd.solver import BatchSolver, status_from_bytes

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the solver has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_dist = os.environ.get("NMPC_BENCH_FORCE_DIST") == "1"      # exercise the RCCL path on a single GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = named_config(args.config)
    B = args.batch
    kw = dict(synthetic_circles=(args.config in ("cfg3", "nobs50")), random_dyn=(args.config in ("cfg4", "smooth_velocity")))
    routes = None
    if args.routes > 0:                                                   # "randomized start/goal" (BASELINE.json configs[1])
        from mpc_trajectory_generator_amd.frontend import random_routes
        routes = random_routes(cfg, args.scene, args.routes, seed=1000 + rank)
    P_host = synthetic_batch(cfg, args.scene, B, seed=rank, routes=routes, **kw)   # timing seeds 0..R-1
    solver = BatchSolver(cfg, max_batch=B, device=local)
    d_p = torch.from_numpy(P_host).to(dev)
    d_u = torch.zeros(B, cfg.n_u, dtype=torch.float64, device=dev)
    d_y = torch.zeros(B, cfg.n1, dtype=torch.float64, device=dev)
    d_st = torch.zeros(B, 72, dtype=torch.uint8, device=dev)
    use_dist = dist is not None
    d_gather = torch.empty(world * B, cfg.n_u, dtype=torch.float64, device=dev) if use_dist else None

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    d_y0 = torch.zeros(B, cfg.n1, dtype=torch.float64, device=dev) if args.warm else None

    def step(i=None):
        if args.warm:
            d_y0.copy_(d_y)                           # warm start: previous solution (in place) and multipliers
        else:
            d_u.zero_()                               # cold start: u0 = 0 (the solver works in place)
        if i is not None:
            ev[i][0].record()
        solver.solve_device(d_p, d_u, d_y0, None, d_y, d_st)
        if i is not None:
            ev[i][1].record()
        if use_dist:
            dist.all_gather_into_tensor(d_gather, d_u)          # result gather over xGMI (RCCL)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # extra, outside the contract's timed region: the same K steps with TWO batches in flight (two handles,
    # two streams, own result buffers) -- the tail of one batch overlaps the bulk of the next, which is how a
    # service that receives batch after batch would run the solver.  Reported under "pipelined", never as `value`.
    pipelined = None
    if not use_dist and not args.warm and args.steps >= 2 and not args.no_pipelined:
        solver2 = BatchSolver(cfg, max_batch=B, device=local)
        bufs = [(solver, d_u, d_y, d_st, torch.cuda.Stream(dev)),
                (solver2, torch.zeros_like(d_u), torch.zeros_like(d_y), torch.zeros_like(d_st), torch.cuda.Stream(dev))]
        fence()
        tp = time.perf_counter()
        for i in range(args.steps):
            sv, bu, by, bs, strm = bufs[i & 1]
            with torch.cuda.stream(strm):
                bu.zero_()
                sv.solve_device(d_p, bu, None, None, by, bs)
        fence()
        tp = time.perf_counter() - tp
        assert torch.equal(bufs[0][1], bufs[1][1])                         # both handles solved the same batch
        pipelined = {"inflight": 2, "value": B * args.steps / tp, "unit": "solves/s", "ms_per_step": 1e3 * tp / args.steps}
        solver2.close()

    st = status_from_bytes(d_st)
    u_gpu = d_u.cpu().numpy()
    y_gpu = d_y.cpu().numpy()
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    f_fwd, f_bwd, f_iter = flop_model(cfg)
    flops = float(st["num_cost_evals"].astype(np.float64).sum() * f_fwd
                  + st["num_grad_evals"].astype(np.float64).sum() * (f_fwd + f_bwd)
                  + st["num_inner_iterations"].astype(np.float64).sum() * f_iter)
    bytes_alg = float(B * (8 * (cfg.n_p + 2 * cfg.n_u + cfg.n1) + 72))
    stats = np.array([st["num_inner_iterations"].sum(), st["num_outer_iterations"].sum(),
                      (st["exit_status"] == 0).sum(), B], dtype=np.float64)
    if use_dist:
        ts_ = torch.from_numpy(stats).to(dev)
        dist.all_reduce(ts_)
        stats = ts_.cpu().numpy()

    if use_dist and world > 1:
        assert torch.equal(d_gather[rank * B:(rank + 1) * B], d_u)      # own shard landed in its slot
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    out = {
        "metric": "nmpc_solves_per_sec", "value": world * B * args.steps / elapsed, "unit": "solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.config}: default.yaml-shaped NMPC, N_hor={cfg.N_hor}, Nobs={cfg.Nobs}, "
                               f"Ndynobs={cfg.Ndynobs}, scene {args.scene} ({args.routes or 1} route(s): "
                               f"{'randomised start/goal planned by the visibility-graph front-end' if args.routes else 'scene start->end'}), "
                               f"batch={B}/GPU, {'WARM start (previous solution and multipliers, c0=1)' if args.warm else 'cold start (u0=0, y0=0, c0=1)'}, "
                               f"tol 1e-4, caps inner {solver.opts.max_inner}/outer {solver.opts.max_outer}",
                   "batch_per_gpu": B, "n_u": cfg.n_u, "n_p": cfg.n_p, "parallelism": f"instance-sharded x{world}"},
        "mean_inner_iters": stats[0] / stats[3], "mean_outer_iters": stats[1] / stats[3],
        "converged_frac": stats[2] / stats[3],
        "pipelined": pipelined,
        "p50_inner_iters": float(np.median(st["num_inner_iterations"])),
        "p99_inner_iters": float(np.percentile(st["num_inner_iterations"], 99)),
        "max_inner_iters": int(st["num_inner_iterations"].max()),
        # compute-bound, priced against the dense f64 peak (MI355X: vector f64 = f64 MFMA = 78.6 TFLOP/s);
        # the kernel issues no MFMA -- "bound_detail" says what actually limits it
        "roofline": {"bound": "mfma", "bound_detail": "valu_f64", "achieved": flops / (kern_ms * 1e-3) / 1e12, "peak": PEAK_F64_VALU_TFLOPS,
                     "unit": "TFLOP/s", "frac": flops / (kern_ms * 1e-3) / 1e12 / PEAK_F64_VALU_TFLOPS,
                     "traffic": PMC_TRAFFIC_BYTES.get(args.config) if (B == 8192 and args.routes == 32 and not args.warm) else None,
                     "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/r01)",
                     "kernel": solver.kernel_name, "kernel_ms": kern_ms,
                     "flops_per_launch": flops,
                     "note": "f64 vector-ALU issue bounds this kernel, not HBM or MFMA (SURVEY.md section 8d); "
                             "MI355X f64 MFMA dense peak is the same 78.6 TFLOP/s"},
        "roofline_hbm": {"bound": "hbm", "achieved": bytes_alg / (kern_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                         "unit": "GB/s", "frac": bytes_alg / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                         "traffic": PMC_TRAFFIC_BYTES.get(args.config) if (B == 8192 and args.routes == 32) else None, "bytes_per_launch": bytes_alg,
                         "traffic_source": "profiles/r01/pmc_fetch.csv + pmc_write.csv (separate --pmc passes)"},
    }

    if not args.no_cpu_baseline and not args.warm:
        # CPU leg: the oracle on the host cores, bounded sample of the same batch (rank 0, any N)
        from oracle import Oracle
        orc = Oracle(cfg.N_hor, cfg.Nobs, cfg.Ndynobs, cfg.ts, cfg.lin_vel_min, cfg.lin_vel_max, cfg.ang_vel_max,
                     cfg.lin_acc_min, cfg.lin_acc_max, cfg.ang_acc_max)
        cores = os.cpu_count() or 1
        n0 = min(B, 4 * cores)
        t = time.perf_counter()
        orc.solve_batch(P_host[:n0], threads=cores)
        rate = n0 / (time.perf_counter() - t)
        n = int(min(B, max(n0, rate * 12.0)))                    # about 12 s of host work
        t = time.perf_counter()
        uo, yo, sto = orc.solve_batch(P_host[:n], threads=cores)
        dt = time.perf_counter() - t
        same = bool(np.array_equal(uo, u_gpu[:n]) and np.array_equal(yo, y_gpu[:n])
                    and np.array_equal(sto["num_inner_iterations"], st["num_inner_iterations"][:n])
                    and np.array_equal(sto["exit_status"], st["exit_status"][:n]))
        out["cpu_baseline"] = {"value": n / dt, "unit": "solves/s", "cores": cores, "kind": "port",
                               "sample": f"first {n} instances of the rank-0 batch, {cores} host threads, "
                                         f"{dt:.1f} s; oracle/nmpc_oracle.c (restatement; OpEn not buildable)",
                               "mean_inner_iters": float(sto["num_inner_iterations"].mean()),
                               "gpu_bitwise_equal_on_sample": same}
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
