"""Multi-GPU plumbing for batches of independent warps: frames shard across ranks, the only exchange
is a one-off broadcast of the shared source texture.  torch.distributed is used as transport only (backend "nccl" is
RCCL on ROCm; "gloo" in the CPU tests).  The C-ABI equivalent for hosts without torch (Node) is hg_multi_* in
include/hgwarp.h (peer copies over xGMI from one process)."""


def shard_frames(n_frames, rank, world):
    """Contiguous block of frame indices for `rank`: sizes differ by at most one, every frame is owned exactly once."""
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def broadcast_source(img_t, rank, world, dist, src=0, verify=False):
    """Shared source texture, rank `src` -> all ranks: scatter 1/N to every peer, then all-gather.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring/tree broadcast is bound by one link, while this way each
    of the root's links carries 1/N of the image and the all-gather uses all links at once.
    img_t: uint8 tensor of identical shape on every rank (content only meaningful on `src`).  Returns the full image.
    Nothing but the two collectives runs here (this is what bench.py times as `broadcast_ms`): when the byte count divides
    by `world` (every RGBA image whose pixel count does) the root scatters views of its own tensor and the all-gather
    lands directly in the result; `verify` (off by default) adds verify_replicas()."""
    if world == 1:
        return img_t
    return scatter_allgather(img_t, rank, world, dist, src, verify)


def scatter_allgather(img_t, rank, world, dist, src=0, verify=False):
    """The two collectives of broadcast_source(), for any world size (world 1 included: the RCCL smoke test on a 1-GPU box)."""
    import torch
    flat = img_t.reshape(-1)
    n = flat.numel()
    chunk = (n + world - 1) // world
    exact = chunk * world == n
    if exact:
        full = flat if rank == src else torch.empty_like(flat)
    else:                                                   # ragged tail: pad (one extra copy on the root)
        full = torch.zeros(chunk * world, dtype=flat.dtype, device=flat.device)
        if rank == src:
            full[:n] = flat
    mine = torch.empty(chunk, dtype=flat.dtype, device=flat.device)
    dist.scatter(mine, list(full.view(world, chunk).unbind(0)) if rank == src else None, src=src)
    out = torch.empty(chunk * world, dtype=flat.dtype, device=flat.device)
    try:
        dist.all_gather_into_tensor(out, mine)
    except (RuntimeError, NotImplementedError, AttributeError):
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        out = torch.cat(parts)
    out = out[:n].view(img_t.shape)
    if verify:
        verify_replicas(out, dist)
    return out


def verify_replicas(t, dist):
    """Raises unless every rank holds the same bytes in `t` (position-weighted 64-bit checksum compared across ranks)."""
    import torch
    flat = t.reshape(-1)
    pad = (-flat.numel()) % 8
    if pad:
        flat = torch.cat([flat, flat.new_zeros(pad)])
    w = flat.view(torch.int64)
    idx = torch.arange(w.numel(), dtype=torch.int64, device=w.device)
    s = (w * (2 * idx + 1)).sum().reshape(1)                # position-weighted, wraps mod 2^64: fine for a checksum
    lo, hi = s.clone(), s.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if int(lo.item()) != int(hi.item()):
        raise RuntimeError("source broadcast: ranks hold different bytes")
