"""Multi-GPU plumbing for batches of independent warps: frames shard across ranks, the only exchange
is a one-off broadcast of the shared source texture.  torch.distributed is used as transport only (backend "nccl" is
RCCL on ROCm; "gloo" in the CPU tests)."""


def shard_frames(n_frames, rank, world):
    """Contiguous block of frame indices for `rank`: sizes differ by at most one, every frame is owned exactly once."""
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def broadcast_source(img_t, rank, world, dist, src=0, verify=True):
    """Shared source texture, rank `src` -> all ranks: scatter 1/N to every peer, then all-gather.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring/tree broadcast is bound by one link, while this way each
    of the root's links carries 1/N of the image and the all-gather uses all links at once.
    img_t: uint8 tensor of identical shape on every rank (content only meaningful on `src`).  Returns the full image."""
    import torch
    if world == 1:
        return img_t
    flat = img_t.reshape(-1)
    n = flat.numel()
    chunk = (n + world - 1) // world
    padded = torch.zeros(chunk * world, dtype=flat.dtype, device=flat.device)
    if rank == src:
        padded[:n] = flat
    mine = torch.empty(chunk, dtype=flat.dtype, device=flat.device)
    dist.scatter(mine, [c.contiguous() for c in padded.view(world, chunk).unbind(0)] if rank == src else None, src=src)
    try:
        dist.all_gather_into_tensor(padded, mine)
    except (RuntimeError, NotImplementedError, AttributeError):
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        padded = torch.cat(parts)
    out = padded[:n].view(img_t.shape).contiguous()
    if verify:
        # every rank must now hold the same bytes: compare a checksum across ranks (all ranks see the same min/max, so
        # they take the same branch) and fall back to the library's plain broadcast if the scatter path disagreed
        s = out.sum(dtype=torch.int64).reshape(1)
        lo, hi = s.clone(), s.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if int(lo.item()) != int(hi.item()):
            out = img_t.contiguous()
            dist.broadcast(out, src=src)
    return out
