#!/usr/bin/env python3
"""bench.py -- NMPC solves/s of the batched HIP solver on BASELINE.json's headline configuration.

One "step" = one cold-start solve (u0 = 0, y0 = 0, c0 = 1) of one batch of B = 8192 independent NMPC
problems per GPU (default.yaml, N = 20, scene 11, 10 circle slots; BASELINE.json configs[1]),
inputs already resident in HBM.  N > 1 GPUs: one process per GPU (torch.distributed, RCCL), each
rank solves its own 8192-instance shard (weak scaling, no data-path collective) and solutions,
multipliers and status structs are gathered over xGMI inside the step
(mpc_trajectory_generator_amd/dist.py: the same pack / gather / unpack the gloo test exercises).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg1|cfg2|cfg3|cfg4] [--batch B]

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment re-launches itself under
``python -m torch.distributed.run`` with N ranks on 127.0.0.1 (and fails loudly when fewer than N GPUs
are visible); launched by torchrun it checks that ``--gpus`` equals WORLD_SIZE.

Prints ONE JSON line (rank 0).  Besides the driver's contract it carries
  roofline      the bound that actually limits the kernel: f64 vector-ALU issue (78.6 TFLOP/s; the
                f64 MFMA dense peak of MI355X is the same number), algorithmic flops from the
                solver's own evaluation counters / the kernel's HIP-event time
  roofline_hbm  algorithmic HBM bytes per launch / the same time, against 8 TB/s (expected << 1 %)
  cpu_baseline  the CPU oracle (oracle/, kind "port": OpEn itself cannot be built here) timed on the
                host cores on a bounded sample of the same batch (dynamic work queue), with the
                single-thread rate beside it, and checked bit-for-bit against the GPU result
  seeds         the same batch recipe with seeds 1 and 2 (beside the timed seed 0): mean and range
  warm_start    the same batch re-solved from the previous solution and multipliers
  converged     throughput and iteration counts of the converged instances alone
  solver_variant  the restatement switches in force
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F64_VALU_TFLOPS = 78.6      # MI355X vector f64 (= f64 MFMA dense peak); MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def usable_cores():
    """Host threads this process may actually run on: the affinity mask, capped by the cgroup CPU quota (a GPU lease
    is typically a slice of the node: os.cpu_count() reports the node, not the lease)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


WORKLOADS = {   # what each BASELINE.json configuration is made of (mpc_trajectory_generator_amd/config.py named_config)
    "cfg1": "configs/default.yaml",
    "cfg2": "configs/default.yaml + jconf_3.yaml's lin_acc_penalty=100 with N_hor overridden to 40",
    "cfg3": "configs/default.yaml with Nobs overridden to 50, all slots filled from a synthetic random-polygon circle field",
    "cfg4": "configs/smooth_velocity.yaml's weights/bounds overlaid on default.yaml (N_hor=20), three random moving ellipses per instance",
}
POINTS_PER_PASS = {"nmpc_solve_hyb_kernel": 3, "nmpc_solve_hyb2_kernel": 3}


def flop_model(cfg):
    """Algorithmic flops (fma = 2, everything else incl. sin/cos/div = 1), DESIGN.md section 6."""
    N, Nobs, Ndyn = cfg.N_hor, cfg.Nobs, cfg.Ndynobs
    f_fwd = N * (28 + 7 * Nobs + 15 * Ndyn + 21 * (N - 1)) + 14 * N
    f_bwd = N * (88 + 9 * Nobs + 16 * Ndyn)                                  # adjoint sweep, as implemented
    f_iter = 70 * cfg.n_u                                                    # L-BFGS two-loop + PANOC vector ops
    return f_fwd, f_bwd, f_iter


def build_report(kernel_name):
    """How the library that ran was built (csrc/build_info.json, written by _lib.build_library): scheduler flags, the code-generation check's
    verdict, and registers / LDS / scratch / spills of the launched kernel.  A library of another origin (NMPC_LIB_PATH) reports that."""
    from mpc_trajectory_generator_amd import _lib
    if os.environ.get("NMPC_LIB_PATH"):
        return {"library": os.environ["NMPC_LIB_PATH"], "note": "not the shipped build"}
    info = _lib.build_info()
    short = kernel_name.split("<")[0]
    tmpl = kernel_name.split("<")[1].rstrip(">") if "<" in kernel_name else ""
    res = {k: v for k, v in info.get("resources", {}).items() if short in k and (not tmpl or tmpl in k or tmpl.isdigit())}
    chk = info.get("codegen_check", {})
    return {"flags": " ".join(info.get("flags") or []), "source_hash": info.get("source_hash"), "stale": bool(info.get("stale")),
            "experiments": bool(_lib.load_library().nmpc_experiments_build()),      # False: the shipped library reads no environment knob
            "codegen_check": {k: chk.get(k) for k in ("ok", "sched_changed", "sched_latent", "exec_hits", "exec_restores")},
            "kernel_resources": next(iter(res.values()), None), "dynamic_lds_note": "LDS is dynamic: one slice per wave, DESIGN.md section 3"}


def pmc_traffic(kernel_name, config, B, routes):
    """HBM bytes per launch from the rocprofv3 PMC passes, looked up in profiles/*/traffic.json by kernel
    name + hash of the kernel sources + workload: a figure measured on another version of the kernels
    (or another workload) is never quoted -- the entry is then null."""
    import glob
    from mpc_trajectory_generator_amd import _lib
    want = {"kernel": kernel_name, "source_hash": _lib.source_hash(), "config": config, "batch": B, "routes": routes}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic.json")), reverse=True):
        try:
            entries = json.load(open(path))
        except (OSError, ValueError):
            continue
        for e in entries if isinstance(entries, list) else [entries]:
            if all(e.get(k) == v for k, v in want.items()):
                return float(e["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT)
    return None, None


def pmc_valu(kernel_name, config):
    """Vector instructions per evaluation pass and the vector ALU's busy share from the SQ counter pass (profiles/*/valu.json), keyed like the
    PMC traffic by kernel + source hash + config; None if not measured for these sources."""
    import glob
    from mpc_trajectory_generator_amd import _lib
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "valu.json")), reverse=True):
        try:
            entries = json.load(open(path))
        except (OSError, ValueError):
            continue
        for e in entries:
            if e.get("kernel") == kernel_name and e.get("config") == config and e.get("source_hash") == _lib.source_hash():
                return e, os.path.relpath(path, ROOT)
    return None, None


def dynamic_mix(kernel_name, config):
    """Executed instructions per evaluation pass by class (scripts/bbcount.py: basic-block counters in the compiler's own assembly), from
    profiles/*/dynamic_mix.json, keyed by kernel + source hash + config; None if not measured for these sources."""
    import glob
    from mpc_trajectory_generator_amd import _lib
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "dynamic_mix.json")), reverse=True):
        try:
            entries = json.load(open(path))
        except (OSError, ValueError):
            continue
        for e in entries if isinstance(entries, list) else [entries]:
            if e.get("kernel") == kernel_name and e.get("config") == config and e.get("source_hash") == _lib.source_hash() and e.get("instance") is None:
                return {k: e.get(k) for k in ("per_pass", "valu_per_pass", "f64_share_of_valu", "all_instructions_per_pass")}, os.path.relpath(path, ROOT)
    return None, None


def scan_shares(kernel_name, config):
    """Share of the evaluations in which the exact certificates of eval_psi fell back to the full cross-track scan / ran the obstacle
    activity scan (counters of a -DNMPC_WIN_STATS build, scripts/win_stats.py -> profiles/*/scan_shares.json), keyed like the PMC traffic
    by kernel + source hash + config; None if not measured for these sources."""
    import glob
    from mpc_trajectory_generator_amd import _lib
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "scan_shares.json")), reverse=True):
        try:
            entries = json.load(open(path))
        except (OSError, ValueError):
            continue
        for e in entries:
            if e.get("kernel") == kernel_name and e.get("config") == config and e.get("source_hash") == _lib.source_hash():
                return e, os.path.relpath(path, ROOT)
    return None, None


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: launch N ranks of this script, one per GPU."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not (have >= 1 and os.environ.get("NMPC_BENCH_SHARED_GPU") == "1"):
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node; refusing to "
                         "report a multi-GPU figure from fewer devices")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


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
    ap.add_argument("--no-extras", action="store_true",
                    help="skip everything outside the contract's timed region (other seeds, warm start, pipelined, CPU leg)")
    ap.add_argument("--warm", action="store_true",
                    help="warm start as the TIMED workload: every step starts from the previous step's solution and "
                         "multipliers (SURVEY.md section 8d second timing); the headline metric is the cold start")
    ap.add_argument("--budget", type=int, default=0,
                    help="max_total_inner: deterministic stand-in for the reference's max_duration (0 = off, the headline)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT",
                    help="solver option / restatement switch (include/nmpc_solver.h), e.g. --opt akkt_gradient=1 --opt ls_failure=1; "
                         "the line then reports that variant under `solver_variant` (the headline uses the defaults)")
    args = ap.parse_args()
    if args.no_extras:
        args.no_cpu_baseline = args.no_pipelined = True

    launched = "WORLD_SIZE" in os.environ
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not launched and args.gpus > 1:
        respawn_under_torchrun(args)

    import torch
    from mpc_trajectory_generator_amd import _lib, named_config
    from mpc_trajectory_generator_amd import dist as shard
    from mpc_trajectory_generator_amd.harness import synthetic_batch
    from mpc_trajectory_generator_amd.solver import BatchSolver, status_from_bytes

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # functional check of the N > 1 path on a box with ONE GPU (tests/test_gpu_dist.py): every rank on device 0, the gather over gloo -- RCCL
    # refuses two ranks on one device.  The line is marked `shared_gpu` and is not a measurement.
    shared_gpu = os.environ.get("NMPC_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the solver has no CPU path")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_dist = os.environ.get("NMPC_BENCH_FORCE_DIST") == "1"      # exercise the RCCL path on a single GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    use_dist = dist is not None

    # rank 0 makes sure the library is built, the others wait for it (build_library also holds a file lock)
    if rank == 0:
        _lib.build_library()
    if use_dist:
        dist.barrier()

    cfg = named_config(args.config)
    B = args.batch
    kw = dict(synthetic_circles=(args.config in ("cfg3", "nobs50")), random_dyn=(args.config in ("cfg4", "smooth_velocity")))

    def make_batch(seed):
        routes = None
        if args.routes > 0:                                               # "randomized start/goal" (BASELINE.json configs[1])
            from mpc_trajectory_generator_amd.frontend import random_routes
            routes = random_routes(cfg, args.scene, args.routes, seed=1000 + seed)
        return synthetic_batch(cfg, args.scene, B, seed=seed, routes=routes, **kw)

    P_host = make_batch(rank)                                             # timing seeds 0..R-1
    opts = {"max_total_inner": args.budget} if args.budget > 0 else {}
    for kv in args.opt:
        k, v = kv.split("=")
        opts[k] = int(v)
    solver = BatchSolver(cfg, max_batch=B, device=local, **opts)
    d_p = torch.from_numpy(P_host).to(dev)
    d_u = torch.zeros(B, cfg.n_u, dtype=torch.float64, device=dev)
    d_y = torch.zeros(B, cfg.n1, dtype=torch.float64, device=dev)
    d_st = torch.zeros(B, 72, dtype=torch.uint8, device=dev)
    d_payload = torch.zeros(B, shard.payload_cols(cfg.n_u, cfg.n1), dtype=torch.float64, device=dev) if use_dist else None
    d_gather = torch.empty(world * B, d_payload.shape[1], dtype=torch.float64, device=dev) if use_dist else None

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    d_y0 = torch.zeros(B, cfg.n1, dtype=torch.float64, device=dev)

    def step(i=None, warm=args.warm, p=d_p):
        if warm:
            d_y0.copy_(d_y)                           # warm start: previous solution (in place) and multipliers
        else:
            d_u.zero_()                               # cold start: u0 = 0 (the solver works in place)
        if i is not None:
            ev[i][0].record()
        solver.solve_device(p, d_u, d_y0 if warm else None, None, d_y, d_st)
        if i is not None:
            ev[i][1].record()
        if use_dist:                                  # result gather over xGMI (RCCL): u | y | status, one collective
            shard.pack_results(d_payload, d_u, d_y, d_st)
            shard.gather_shards(d_payload, out=d_gather)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(n, **kws):
        fence()
        t = time.perf_counter()
        for _ in range(n):
            step(**kws)
        fence()
        return (time.perf_counter() - t) / n

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

    st = status_from_bytes(d_st).copy()
    u_gpu = d_u.cpu().numpy()
    y_gpu = d_y.cpu().numpy()
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    gather_ok = None
    if use_dist:
        # every rank's shard landed in its slot of the gathered payload, bit for bit
        Ug, Yg, stg = shard.unpack_results(d_gather, world * B, world, cfg.n_u, cfg.n1, _lib.STATUS_DTYPE)
        gather_ok = bool(np.array_equal(Ug[rank * B:(rank + 1) * B], u_gpu) and np.array_equal(Yg[rank * B:(rank + 1) * B], y_gpu)
                         and stg[rank * B:(rank + 1) * B].tobytes() == st.tobytes())
        assert gather_ok, "gathered payload does not hold this rank's results"
        assert all((stg["num_inner_iterations"][r * B:(r + 1) * B] > 0).any() for r in range(world)), "a rank's slot is empty"

    # per-rank report (device, shard, kernel ms) gathered to rank 0, and the gather alone (pack + all_gather), timed
    # by events outside the timed region: a wrong rank -> device mapping or a slow link shows up in the line
    ranks_report = gather_alone = None
    if use_dist:
        gev = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            shard.pack_results(d_payload, d_u, d_y, d_st)
            shard.gather_shards(d_payload, out=d_gather)
            e1.record()
            gev.append((e0, e1))
        fence()
        gms = float(np.mean([a.elapsed_time(b) for a, b in gev[1:]]))
        props = torch.cuda.get_device_properties(local)
        mine = {"rank": rank, "local_rank": local, "device": torch.cuda.current_device(), "name": props.name,
                "pci_bus_id": getattr(props, "pci_bus_id", None), "shard": [rank * B, (rank + 1) * B],
                "kernel_ms": kern_ms, "gather_ms": gms, "host": socket.gethostname()}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        ranks_report = allr
        gather_alone = {"ms": max(r["gather_ms"] for r in allr), "bytes_per_rank": int(d_payload.numel() * 8),
                        "what": "pack_results + one all_gather_into_tensor, HIP events, max over ranks"}
        devs = [r["device"] for r in allr if r["host"] == mine["host"]]
        assert shared_gpu or len(set(devs)) == len(devs), f"two ranks share a device: {allr}"

    f_fwd, f_bwd, f_iter = flop_model(cfg)

    def flops_of(s):
        return float(s["num_cost_evals"].astype(np.float64).sum() * f_fwd
                     + s["num_grad_evals"].astype(np.float64).sum() * (f_fwd + f_bwd)
                     + s["num_inner_iterations"].astype(np.float64).sum() * f_iter)

    flops = flops_of(st)
    # SURVEY.md section 8(d)'s literal formula beside it: n_cost F_cost + n_grad 4 F_cost + iterations 46 n_u (its "reverse mode = 3 x forward" guess
    # for the adjoint; the as-implemented adjoint above is far cheaper, so this figure is the larger one)
    flops_survey = float(st["num_cost_evals"].astype(np.float64).sum() * f_fwd + st["num_grad_evals"].astype(np.float64).sum() * 4.0 * f_fwd
                         + st["num_inner_iterations"].astype(np.float64).sum() * 46.0 * cfg.n_u)
    # what the ALUs execute of that: the windowed cross-track search measures three segments instead of N - 1, the obstacle certificate
    # skips the activity scan -- both exact, both fall back to the full work in a measured share of the evaluations
    shares, shares_src = scan_shares(solver.kernel_name, args.config)
    executed_frac = None
    if shares:
        N_, No_, Nd_ = cfg.N_hor, cfg.Nobs, cfg.Ndynobs
        sw, so = shares["window_full_scan_share"], shares["obstacle_scan_share"]
        f_fwd_x = N_ * (28 + (7 * No_ + 15 * Nd_) * so + 21 * ((N_ - 1) * sw + 3 * (1 - sw))) + 14 * N_
        n_ev = float(st["num_cost_evals"].astype(np.float64).sum() + st["num_grad_evals"].astype(np.float64).sum())
        executed_frac = (flops - n_ev * (f_fwd - f_fwd_x)) / flops
    else:
        print(f"bench.py: no certificate shares for these kernel sources under profiles/*/scan_shares.json -- roofline.executed_frac is null "
              f"(scripts/win_stats.py {args.config} measures them)", file=sys.stderr)
    valu_pmc, valu_src = pmc_valu(solver.kernel_name, args.config)
    mix, mix_src = dynamic_mix(solver.kernel_name, args.config)
    bytes_alg = float(B * (8 * (cfg.n_p + 2 * cfg.n_u + cfg.n1) + 72))
    conv = st["exit_status"] == 0
    stats = np.array([st["num_inner_iterations"].sum(), st["num_outer_iterations"].sum(), conv.sum(), B,
                      st["num_inner_iterations"][conv].sum()], dtype=np.float64)
    if use_dist:
        ts_ = torch.from_numpy(stats).to(dev)
        dist.all_reduce(ts_)
        stats = ts_.cpu().numpy()

    # ------------------------------------------------------------------ extras, all outside the timed region
    extras_ok = not use_dist and not args.warm and not args.no_extras and args.budget == 0 and not args.opt
    seeds = warm = pipelined = None
    if extras_ok:
        # (a) seeds 1 and 2 of the same recipe: the launch-order heuristic must not be fit to seed 0
        per_seed = {"0": 1e3 * elapsed / args.steps}
        for sd in (1, 2):
            d_ps = torch.from_numpy(make_batch(sd)).to(dev)
            step(p=d_ps)
            per_seed[str(sd)] = 1e3 * timed(2, p=d_ps)
            del d_ps
        ms = np.array(list(per_seed.values()))
        seeds = {"ms_per_step": per_seed, "mean_ms": float(ms.mean()), "min_ms": float(ms.min()), "max_ms": float(ms.max()),
                 "mean_solves_per_s": float(B / (ms.mean() * 1e-3)),
                 "note": "seed 0 is the timed workload (`value`); seeds 1, 2: one untimed + two timed steps each"}
        # (b) warm start: the batch re-solved from ITS cold-start solution and multipliers (c0 = 1), every repetition from
        # the same starting point (the kernel alone is timed, with events, so that restoring the start is not in the figure)
        step(warm=False)
        fence()
        u_cold, y_cold = d_u.clone(), d_y.clone()
        wev = []
        for _ in range(3):
            d_u.copy_(u_cold)
            d_y0.copy_(y_cold)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            solver.solve_device(d_p, d_u, d_y0, None, d_y, d_st)
            e1.record()
            wev.append((e0, e1))
        fence()
        wms = float(np.mean([a.elapsed_time(b) for a, b in wev[1:]]))
        stw = status_from_bytes(d_st)
        warm = {"value": B / (wms * 1e-3), "unit": "solves/s", "ms_per_step": wms,
                "mean_inner_iters": float(stw["num_inner_iterations"].mean()),
                "converged_frac": float((stw["exit_status"] == 0).mean()),
                "roofline_frac": flops_of(stw) / (wms * 1e-3) / 1e12 / PEAK_F64_VALU_TFLOPS,
                "note": "each repetition restarts from the cold-start solution and multipliers of the same batch"}
        del u_cold, y_cold
        step(warm=False)                                                   # leave the cold-start results in the buffers
        fence()
    # (b2) the other values of the one restatement switch that moves the headline by a large factor,
    # timed by HIP events on the same batch, one warm-up + one timed step each
    variants = None
    if extras_ok:
        variants = {}
        for val in (0, 1, 2):
            if val == solver.opts.akkt_gradient:
                continue
            sv = BatchSolver(cfg, max_batch=B, device=local, akkt_gradient=val)
            vev = []
            for _ in range(2):
                d_u.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                sv.solve_device(d_p, d_u, None, None, d_y, d_st)
                e1.record()
                vev.append((e0, e1))
            fence()
            vms = float(vev[1][0].elapsed_time(vev[1][1]))
            stv = status_from_bytes(d_st)
            variants[f"akkt_gradient={val} ({sv.variant['akkt_gradient']})"] = {
                "value": B / (vms * 1e-3), "unit": "solves/s", "ms_per_step": vms,
                "mean_inner_iters": float(stv["num_inner_iterations"].mean()),
                "converged_frac": float((stv["exit_status"] == 0).mean()),
                "roofline_frac": flops_of(stv) / (vms * 1e-3) / 1e12 / PEAK_F64_VALU_TFLOPS}
            sv.close()
        step(warm=False)                                                   # leave the default variant's results in the buffers
        fence()
    # (c) the same K steps with TWO batches in flight (two handles, two streams, own result buffers): the tail
    # of one batch overlaps the bulk of the next, which is how a service that receives batch after batch would
    # run the solver.  Reported under "pipelined", never as `value`.
    if extras_ok and args.steps >= 2 and not args.no_pipelined:
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

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    traffic, traffic_src = pmc_traffic(solver.kernel_name, args.config, B, args.routes) if not args.warm else (None, None)
    value = world * B * args.steps / elapsed
    out = {
        "metric": "nmpc_solves_per_sec", "value": value, "unit": "solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.config}: {WORKLOADS.get(args.config, args.config)}; N_hor={cfg.N_hor}, Nobs={cfg.Nobs}, "
                               f"Ndynobs={cfg.Ndynobs}, scene {args.scene} ({args.routes or 1} route(s): "
                               f"{'randomised start/goal planned by the visibility-graph front-end' if args.routes else 'scene start->end'}), "
                               f"batch={B}/GPU, {'WARM start (previous solution and multipliers, c0=1)' if args.warm else 'cold start (u0=0, y0=0, c0=1)'}, "
                               f"tol 1e-4, caps inner {solver.opts.max_inner}/outer {solver.opts.max_outer}"
                               + (f", inner-iteration budget {args.budget} per solve (NotConvergedOutOfTime beyond it)" if args.budget else ""),
                   "batch_per_gpu": B, "n_u": cfg.n_u, "n_p": cfg.n_p, "parallelism": f"instance-sharded x{world}",
                   "gather": ("one gloo all_gather of u|y|status per step, ALL RANKS ON ONE GPU: functional check only" if shared_gpu else
                              "one RCCL all_gather of u|y|status per step") if use_dist else "none (single GPU)"},
        "solver_variant": solver.variant,
        "build": build_report(solver.kernel_name),
        "mean_inner_iters": stats[0] / stats[3], "mean_outer_iters": stats[1] / stats[3],
        "converged_frac": stats[2] / stats[3],
        # the headline counts every solve, converged or not (a non-converged solve still returns the controls the
        # reference would apply, src/path_generator.py:393-394); the converged ones alone:
        "converged": {"value": value * stats[2] / stats[3], "unit": "converged solves/s",
                      "mean_inner_iters": stats[4] / max(stats[2], 1.0)},
        "gather_checked": gather_ok, "gather_alone": gather_alone, "ranks": ranks_report, "shared_gpu": shared_gpu or None,
        "seeds": seeds, "warm_start": warm, "pipelined": pipelined, "variants": variants,
        "p50_inner_iters": float(np.median(st["num_inner_iterations"])),
        "p99_inner_iters": float(np.percentile(st["num_inner_iterations"], 99)),
        "max_inner_iters": int(st["num_inner_iterations"].max()),
        # the batch ends with its slowest instance: its evaluation passes, and the batch time spread over them
        "critical_instance": {"passes": int(st["reserved"].max()), "kernel_us_per_critical_pass": 1e3 * kern_ms / max(int(st["reserved"].max()), 1),
                              "mean_passes": float(st["reserved"].mean()),
                              "points_per_pass": POINTS_PER_PASS.get(solver.kernel_name.split("<")[0]),
                              "slowest_instance_ms": float(st["solve_time_ms"].max()),
                              "mean_instance_ms": float(st["solve_time_ms"].mean())},
        # compute-bound, priced against the dense f64 peak (MI355X: vector f64 = f64 MFMA = 78.6 TFLOP/s).  `bound` stays inside the bench contract's
        # enum -- "mfma" is its compute-side roofline --, `bound_class` says which unit it is: the f64 vector ALU (the kernel issues no MFMA)
        "roofline": {"bound": "mfma", "bound_class": "valu_f64",
                     "model": "algorithmic flops of the sequential method from the solver's own counters: F_fwd = N(28 + 7 Nobs + 15 Ndyn + 21 (N - 1)) + 14 N "
                              "per evaluation (SURVEY.md App. G), + the as-implemented adjoint N(88 + 9 Nobs + 16 Ndyn) per gradient (NOT the survey's 4 x F_cost guess), "
                              "+ 70 n_u per PANOC iteration; fma = 2",
                     "achieved": flops / (kern_ms * 1e-3) / 1e12, "peak": PEAK_F64_VALU_TFLOPS,
                     "unit": "TFLOP/s", "frac": flops / (kern_ms * 1e-3) / 1e12 / PEAK_F64_VALU_TFLOPS,
                     # the same launch by SURVEY.md section 8(d)'s literal formula (cost + gradient = 4 F_cost, 46 n_u per iteration)
                     "frac_survey_model": flops_survey / (kern_ms * 1e-3) / 1e12 / PEAK_F64_VALU_TFLOPS, "flops_per_launch_survey_model": flops_survey,
                     "executed_frac": executed_frac, "executed_frac_source": shares_src,
                     "executed_frac_note": "share of the credited flops the ALUs execute: the exact windowed cross-track search and the exact obstacle "
                                           "certificate skip the rest (null: certificates not counted for these sources)",
                     # what the ALUs execute of the credited flops, per second and against the peak
                     "achieved_executed": None if executed_frac is None else executed_frac * flops / (kern_ms * 1e-3) / 1e12,
                     "frac_executed": None if executed_frac is None else executed_frac * flops / (kern_ms * 1e-3) / 1e12 / PEAK_F64_VALU_TFLOPS,
                     "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC)", "traffic_source": traffic_src,
                     # vector instructions per evaluation pass and the vector ALU's busy share (rocprofv3 SQ counters), executed instructions per pass by
                     # class (basic-block counters, scripts/bbcount.py): all keyed by the hash of the kernel sources, null when measured for other sources
                     "valu_per_pass": None if not valu_pmc else valu_pmc["valu_per_pass"], "valu_busy": None if not valu_pmc else valu_pmc["valu_busy"],
                     "valu_source": valu_src, "dynamic_mix": mix, "dynamic_mix_source": mix_src,
                     "kernel": solver.kernel_name, "kernel_ms": kern_ms,
                     "flops_per_launch": flops,
                     "note": "instruction issue bounds this kernel -- a wave issues one instruction of any class per slot, 60 % of them vector-ALU, half of "
                             "those f64 arithmetic --, not HBM or MFMA (SURVEY.md section 8d; DESIGN.md section 5.6); MI355X f64 MFMA dense peak is the same 78.6 TFLOP/s"},
        "roofline_hbm": {"bound": "hbm", "achieved": bytes_alg / (kern_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                         "unit": "GB/s", "frac": bytes_alg / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                         "traffic": traffic, "bytes_per_launch": bytes_alg, "traffic_source": traffic_src},
    }

    if not args.no_cpu_baseline and not args.warm and world == 1:
        # CPU leg: the oracle on the host cores, bounded sample of the same batch (rank 0, N = 1 only)
        from oracle import Oracle
        orc = Oracle(cfg.N_hor, cfg.Nobs, cfg.Ndynobs, cfg.ts, cfg.lin_vel_min, cfg.lin_vel_max, cfg.ang_vel_max,
                     cfg.lin_acc_min, cfg.lin_acc_max, cfg.ang_acc_max, **solver.oracle_opts())
        cores, quota = usable_cores()                            # threads = the cores this process may run on
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
        t = time.perf_counter()                                  # one thread: a short probe sizes a sample of about 5 s
        orc.solve_batch(P_host[:16], threads=1)
        r1 = 16 / (time.perf_counter() - t)
        n1t = int(min(n, max(16, r1 * 5.0)))
        t = time.perf_counter()
        orc.solve_batch(P_host[:n1t], threads=1)
        dt1 = time.perf_counter() - t
        out["cpu_baseline"] = {"value": n / dt, "unit": "solves/s", "cores": cores, "threads": cores, "kind": "port",
                               "host_cpu_count": os.cpu_count(), "cgroup_cpu_quota": quota,
                               "speedup_over_one_thread": (n / dt) / (n1t / dt1),
                               "sample": f"first {n} instances of the rank-0 batch, {cores} host threads pulling from a "
                                         f"shared work queue, {dt:.1f} s; oracle/nmpc_oracle.c -O3 (restatement; OpEn not buildable)",
                               "single_thread": {"value": n1t / dt1, "unit": "solves/s", "sample": f"first {n1t} instances, {dt1:.1f} s"},
                               "mean_inner_iters": float(sto["num_inner_iterations"].mean()),
                               "gpu_bitwise_equal_on_sample": same}
    # the JSON line must be the LAST thing on stdout: tear RCCL down first and flush C stdio (RCCL prints its version
    # banner through it, which would otherwise land after this line when stdout is a pipe or a file)
    if use_dist:
        dist.destroy_process_group()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
