"""How much of a batch's time is launch order?  Per-instance pass counts (status.reserved) of a config's batch from the GPU, then a
list-scheduling model (S wave slots, constant microseconds per pass) for: the shipped order (the classifier of nmpc_classify_kernel restated
in numpy), exact longest-first, index order, and 'probe first': every instance runs its first outer iteration, then the rest is ordered by what
that iteration revealed.  usage: python scripts/order_model.py cfgN [seed]   -> JSON (also dumps gpurun_out/passes_<cfg>_<seed>.npz)"""
import heapq, json, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.solver import BatchSolver
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.frontend import random_routes

name = sys.argv[1]; seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = named_config(name)
B = 8192
P = synthetic_batch(cfg, 11, B, seed, routes=random_routes(cfg, 11, 32, seed=1000 + seed), synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
s = BatchSolver(cfg, max_batch=B)
u, y, st = s.solve(P); ms = s.last_batch_ms
u, y, st = s.solve(P); ms = min(ms, s.last_batch_ms)
kernel = s.kernel_name
s.close()
s1 = BatchSolver(cfg, max_batch=B, max_outer=1)
_, _, st1 = s1.solve(P)
s1.close()
passes = st["reserved"].astype(np.int64); p1 = st1["reserved"].astype(np.int64)
np.savez(f"gpurun_out/passes_{name}_{seed}.npz", passes=passes, p1=p1, inner=st["num_inner_iterations"], inner1=st1["num_inner_iterations"],
         f2_1=st1["f2_norm"], dy1=st1["delta_y_norm_over_c"], outer=st["num_outer_iterations"], ms=st["solve_time_ms"])

def levels(P):          # nmpc_classify_kernel (csrc/nmpc_kernels.hip)
    N, nobs, ndyn = cfg.N_hor, cfg.Nobs, cfg.Ndynobs
    ps = P[:, 20 + N: 20 + N + 3 * nobs].reshape(B, nobs, 3)
    pd = P[:, 20 + N + 3 * nobs: 20 + N + 3 * nobs + 5 * ndyn * N].reshape(B, ndyn, N, 5)
    pr = P[:, 20 + N + 3 * nobs + 5 * ndyn * N:].reshape(B, N, 3)
    dth = np.diff(pr[:, :, 2], axis=1); dth = dth - 2 * np.pi * np.rint(dth / (2 * np.pi)); bend = np.abs(dth).sum(1)
    d2 = ((pr[:, :, None, 0] - ps[:, None, :, 0]) ** 2 + (pr[:, :, None, 1] - ps[:, None, :, 1]) ** 2)      # [B, N, nobs]
    r = ps[:, None, :, 2]; real = r > 0
    hard = (real & (d2 < (r + 0.6) ** 2)).any((1, 2)); g = real & (d2 < (r + 0.05) ** 2)
    graze = g.any((1, 2)); early = g[:, : (N + 1) // 2].any((1, 2))
    if ndyn:
        e = pd.transpose(0, 2, 1, 3)      # [B, N, ndyn, 5]
        dd = (pr[:, :, None, 0] - e[..., 0]) ** 2 + (pr[:, :, None, 1] - e[..., 1]) ** 2
        hard |= (dd < (np.maximum(e[..., 2], e[..., 3]) + 0.6) ** 2).any((1, 2)); graze |= (dd < (np.minimum(e[..., 2], e[..., 3]) + 0.05) ** 2).any((1, 2))
    gap = np.abs(P[:, 20] - P[:, 3]) > 1.0
    goal = (pr[:, -1, 0] == pr[:, -2, 0]) & (pr[:, -1, 1] == pr[:, -2, 1])
    return 4 * graze + 4 * early + hard + (bend > 0.05) + gap + goal

def makespan(order, work, S, t0=None):
    h = list(t0) if t0 is not None else [0.0] * S
    heapq.heapify(h)
    end = 0.0
    for i in order:
        t = heapq.heappop(h) + work[i]
        end = max(end, t); heapq.heappush(h, t)
    return end, h

two_stage = "hyb2" in kernel
S = 1024 if two_stage else 2048
us = ms * 1e3 * S / passes.sum() * 0.0 + (5.7 if two_stage else 6.0)      # microseconds per pass of a resident wave
w = passes * us * 1e-3
lv = levels(P)
shipped = np.argsort(-lv, kind="stable")
res = {"cfg": name, "seed": seed, "kernel": kernel, "measured_ms": ms, "slots": S, "us_per_pass": us, "mean_passes": float(passes.mean()), "max_passes": int(passes.max()),
       "perfect_packing_ms": float(w.sum() / S), "longest_alone_ms": float(w.max()),
       "model_shipped_ms": makespan(shipped, w, S)[0], "model_index_ms": makespan(np.arange(B), w, S)[0], "model_lpt_ms": makespan(np.argsort(-w), w, S)[0]}
# probe first: outer iteration 1 of everything (shipped order), then the remainder by a predictor built from what it revealed
w1 = p1 * us * 1e-3; rem = np.maximum(w - w1, 0.0)
end1, heap = makespan(shipped, w1, S)
for label, key in (("remaining_exact", -rem), ("first_outer_passes", -p1.astype(float)), ("f2_then_passes", -(1e6 * (st1["f2_norm"] > 1e-4) + p1))):
    todo = [i for i in np.argsort(key, kind="stable") if rem[i] > 0]
    res["model_probe_first_" + label + "_ms"] = makespan(todo, rem, S, heap)[0]
res["corr_first_outer_passes_vs_remaining"] = float(np.corrcoef(p1, rem)[0, 1])
res["share_of_work_in_first_outer"] = float(w1.sum() / w.sum())
print(json.dumps(res))
