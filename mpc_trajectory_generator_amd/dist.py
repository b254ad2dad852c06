"""Multi-GPU: the batch shards over ranks, nothing else.

Every problem instance is an independent optimisation problem, so N GPUs = N processes (one per
GPU, ``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests), each
solving a contiguous slice of the batch with no data-path collective.  The only communication is
the result gather at the end: ~3.2 MB per rank at 8192 instances -- latency-bound
on one xGMI link, so a plain all_gather is all it takes.
"""
from __future__ import annotations

import numpy as np


def shard_range(B: int, rank: int, world: int):
    """Contiguous split; the first B % world ranks take one extra instance (ragged batches)."""
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def solve_sharded(solve_fn, P, u0=None, y0=None, c0=None, group=None, device=None):
    """Solve rank's slice of ``P`` with ``solve_fn(P, u0, y0, c0) -> (U, Y, status)`` and gather
    everything on every rank.  Returns (U [B, n_u], Y [B, n1], status [B]) in batch order."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    B = P.shape[0]
    lo, hi = shard_range(B, rank, world)
    sl = slice(lo, hi)
    U, Y, st = solve_fn(P[sl], None if u0 is None else u0[sl], None if y0 is None else y0[sl],
                        None if c0 is None else c0[sl])
    n_u, n1 = U.shape[1], Y.shape[1]
    # ragged shards: pad to the largest shard, gather, strip
    cap = -(-B // world)
    payload = np.zeros((cap, n_u + n1 + st.dtype.itemsize // 8))
    payload[:hi - lo, :n_u] = U
    payload[:hi - lo, n_u:n_u + n1] = Y
    payload[:hi - lo, n_u + n1:] = np.frombuffer(st.tobytes(), dtype=np.float64).reshape(hi - lo, -1)
    t = torch.from_numpy(payload)
    if device is not None:
        t = t.to(device)
    out = torch.empty((world * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)          # concatenated along dim 0
    out = out.cpu().numpy().reshape(world, cap, -1)
    Us, Ys, sts = [], [], []
    for r in range(world):
        a, b = shard_range(B, r, world)
        blk = out[r, :b - a]
        Us.append(blk[:, :n_u]), Ys.append(blk[:, n_u:n_u + n1])
        sts.append(np.frombuffer(np.ascontiguousarray(blk[:, n_u + n1:]).tobytes(), dtype=st.dtype))
    return np.concatenate(Us), np.concatenate(Ys), np.concatenate(sts)
