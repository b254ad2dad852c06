"""Multi-GPU: the batch shards over ranks, nothing else.

Every problem instance is an independent optimisation problem, so N GPUs = N processes (one per
GPU, ``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests), each
solving a contiguous slice of the batch with no data-path collective.  The only communication is
the result gather at the end: solutions, multipliers and status structs of a
rank travel as ONE payload row per instance -- 8192 x (40 + 40 + 9) doubles = 5.8 MB per rank --
latency-bound on one xGMI link, so a plain all_gather is all it takes.

``pack_results`` / ``gather_shards`` / ``unpack_results`` are the one code path for this; ``bench.py``
(device tensors, nccl) and ``solve_sharded`` (numpy in / numpy out, any backend) both go through it.
"""
from __future__ import annotations

import numpy as np

STATUS_DOUBLES = 9          # sizeof(nmpc_status) / 8


def shard_range(B: int, rank: int, world: int):
    """Contiguous split; the first B % world ranks take one extra instance (ragged batches)."""
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def payload_cols(n_u: int, n1: int) -> int:
    return n_u + n1 + STATUS_DOUBLES


def pack_results(payload, u, y, status_bytes):
    """Fill rows [0, len(u)) of ``payload`` [cap, n_u + n1 + 9] (float64, same device as the operands)
    with u | y | the 72-byte status structs reinterpreted as 9 doubles.  ``status_bytes`` is the
    uint8 [B, 72] tensor the solver wrote."""
    import torch
    B, n_u, n1 = u.shape[0], u.shape[1], y.shape[1]
    payload[:B, :n_u].copy_(u)
    payload[:B, n_u:n_u + n1].copy_(y)
    payload[:B, n_u + n1:].copy_(status_bytes.view(torch.float64).reshape(B, STATUS_DOUBLES))
    return payload


def gather_shards(payload, out=None, group=None):
    """One all_gather of equally sized shards: [cap, cols] on every rank -> [world * cap, cols] on every
    rank, rank r's shard in rows [r * cap, (r + 1) * cap).  The data stays on its device."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * payload.shape[0], payload.shape[1]), dtype=payload.dtype, device=payload.device)
    dist.all_gather_into_tensor(out, payload, group=group)
    return out


def unpack_results(gathered, B: int, world: int, n_u: int, n1: int, status_dtype):
    """Gathered payload (host or device tensor) -> (U [B, n_u], Y [B, n1], status [B]) in batch order,
    stripping the padding rows of ragged shards."""
    cap = gathered.shape[0] // world
    g = gathered.cpu().numpy().reshape(world, cap, -1)
    Us, Ys, sts = [], [], []
    for r in range(world):
        a, b = shard_range(B, r, world)
        blk = g[r, :b - a]
        Us.append(blk[:, :n_u])
        Ys.append(blk[:, n_u:n_u + n1])
        sts.append(np.frombuffer(np.ascontiguousarray(blk[:, n_u + n1:]).tobytes(), dtype=status_dtype))
    return np.concatenate(Us), np.concatenate(Ys), np.concatenate(sts)


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
    assert st.dtype.itemsize == 8 * STATUS_DOUBLES
    cap = -(-B // world)                       # ragged shards: pad to the largest shard, gather, strip
    payload = torch.zeros((cap, payload_cols(n_u, n1)), dtype=torch.float64, device=device)
    n = hi - lo
    if n:
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(payload.device)      # noqa: E731
        st_bytes = to(np.frombuffer(st.tobytes(), dtype=np.uint8).reshape(n, 8 * STATUS_DOUBLES))
        pack_results(payload, to(U), to(Y), st_bytes)
    out = gather_shards(payload, group=group)
    return unpack_results(out, B, world, n_u, n1, st.dtype)
