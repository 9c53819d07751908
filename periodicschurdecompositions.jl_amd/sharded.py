"""Period-sharded `pschur!` over several engines, one process per GPU (DESIGN.md "Multi-GPU").

Rank g of G owns the contiguous slice of the period `period_slice(p, G, g)` of the Schur vectors: it forms Q_j and
applies the Z role of every bulk update only for those j.  The latency-bound chains (the n p reflector links of the
periodic Hessenberg reduction, the window chases of the QR iteration) and the updates of the factors they read run
on every rank, identically and without any exchange: a hand-off of the chain at slice boundaries would put a
point-to-point latency of several microseconds on each of the chain's microsecond links, and the
chase of every bulge needs the active diagonal block of ALL p factors.  The one collective is the all-gather of the Z
slices at the end (RCCL over xGMI on the GPUs: `backend="nccl"`; gloo in the CPU rehearsal of tests/test_dist_gloo.py).
"""
import numpy as np


def period_slice(p, world, rank):
    """Contiguous slice [lo, hi) of the period owned by `rank` (the rule of psd_ctx::slice in csrc/psd_engine.cpp)."""
    base, rem = divmod(p, world)
    lo = rank * base + min(rank, rem)
    return [lo, lo + base + (1 if rank < rem else 0)]


def pschur_sharded(eng, A, lr="R", dist=None, gather=True, **kw):
    """pschur(A, lr) on a period-sharded engine (host matrices).  `dist`: an initialised torch.distributed module
    (or None for a single process).  Returns the PeriodicSchur; with gather=True every rank ends up with every Z_j."""
    import torch

    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    p = len(A)
    eng.set_shard(rank, world)
    try:
        ps = eng.pschur(A, lr, **kw)
    finally:
        eng.set_shard(0, 1)
    ps.owned = _owned(p, lr, world, rank)
    if world == 1 or not gather or not ps.Z:
        return ps
    n = A[0].shape[0]
    # all-gather of the owned blocks, padded to the largest slice (slices differ by at most one factor)
    per = max(sum(_owned(p, lr, world, r)) for r in range(world))
    mine = [j for j in range(p) if ps.owned[j]]
    # (the element type of the Schur vectors: complex128 for psd_z_pschur — a float64 buffer would drop the imaginary parts)
    tdtype = torch.complex128 if np.iscomplexobj(ps.Z[mine[0]] if mine else ps.Z[0]) else torch.float64
    send = torch.zeros((per, n, n), dtype=tdtype)
    for k, j in enumerate(mine):
        send[k] = torch.from_numpy(np.ascontiguousarray(ps.Z[j].T))  # (column-major block as a row-major tensor)
    recv = [torch.zeros_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    for r in range(world):
        theirs = [j for j in range(p) if _owned(p, lr, world, r)[j]]
        for k, j in enumerate(theirs):
            ps.Z[j] = np.asfortranarray(recv[r][k].numpy().T)
    return ps


def _owned(p, lr, world, rank):
    lo, hi = period_slice(p, world, rank)
    owned = [False] * p
    for j in range(lo, hi):  # internal factor j -> user slot of Z_j (PSD.jl:1078-1092)
        owned[j if (lr in ("R", ":R") or j == 0) else p - j] = True
    return owned


def allgather_z_device(dist, dZ, p, world, rank):
    """Device-resident all-gather of the Z slices of an 'R' decomposition: dZ is the [p, n, n] CUDA tensor every rank
    passed to pschur_dev; on return all p blocks are valid everywhere.  Equal slices go through one
    all_gather_into_tensor (a single RCCL ring all-gather of p n^2 8 / G bytes per rank); ragged ones per block."""
    lo, hi = period_slice(p, world, rank)
    if dist.get_backend() == "gloo":  # (rehearsal on one GPU / on CPUs: gloo gathers host tensors)
        import torch

        per = max(period_slice(p, world, r)[1] - period_slice(p, world, r)[0] for r in range(world))
        send = torch.zeros((per,) + tuple(dZ.shape[1:]), dtype=dZ.dtype)
        send[: hi - lo] = dZ[lo:hi].cpu()
        recv = [torch.zeros_like(send) for _ in range(world)]
        dist.all_gather(recv, send)
        for r in range(world):
            l, h = period_slice(p, world, r)
            if r != rank and h > l:
                dZ[l:h] = recv[r][: h - l].to(dZ.device)
        return
    if p % world == 0:
        dist.all_gather_into_tensor(dZ.view(-1), dZ[lo:hi].reshape(-1).clone())
        return
    for r in range(world):
        l, h = period_slice(p, world, r)
        if h > l:
            dist.broadcast(dZ[l:h], src=r)
