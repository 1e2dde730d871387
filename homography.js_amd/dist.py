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


# ------------------------------------------------------------------ bench.py's N > 1 bookkeeping (pure functions + collectives on
# whatever device the tensors live on: RCCL on the GPU box, gloo in tests/test_dist_cpu.py)

def rank_frame_ids(rank, frames_per_rank):
    """Weak scaling: every rank warps `frames_per_rank` frames of the config's sequence, rank r takes the r-th block."""
    return [rank * frames_per_rank + f for f in range(frames_per_rank)]


def job_frame_ids(scaling, rank, world, frames_per_rank, batch):
    """Which frames of the config's sequence `rank` warps per step.
    weak   (default): `frames_per_rank` frames on every rank, rank r takes the r-th block: per-GPU work is fixed as N grows;
    strong (north_star's wording, "batch = 512 / 64 frames sharded over 8 GPUs"): a FIXED batch of `batch` frames split by
           shard_frames() -- contiguous blocks whose sizes differ by at most one; total work is fixed as N grows."""
    if scaling == "strong":
        return list(shard_frames(batch, rank, world))
    return rank_frame_ids(rank, frames_per_rank)


def end_to_end(ms_per_step, broadcast_ms, pixels_per_step_all):
    """One batch INCLUDING the one-off fan-out of the shared source (paid once per batch, not per step): what a caller who ships a new
    source texture with every batch sees.  Returns (ms per batch, Mpixels/s)."""
    ms = float(ms_per_step) + float(broadcast_ms)
    return ms, (float(pixels_per_step_all) / (ms * 1e-3) / 1e6 if ms > 0 else 0.0)


def gather_rank_info(dist, world, info):
    """Every rank's `info` dict (device name, XCC count, frames it owns ...) on every rank, in rank order."""
    if world == 1:
        return [info]
    out = [None] * world
    dist.all_gather_object(out, info)
    return out


def check_launch(world, rank, local_rank, n_devices):
    """A mis-launch must be loud: more ranks than GPUs on this node, or a local rank without a device, raise."""
    if not (0 <= rank < world):
        raise RuntimeError(f"RANK={rank} outside WORLD_SIZE={world}")
    if world > n_devices:
        raise RuntimeError(f"WORLD_SIZE={world} but this node shows {n_devices} GPU(s): one rank per GPU of ONE node")
    if not (0 <= local_rank < n_devices):
        raise RuntimeError(f"LOCAL_RANK={local_rank} has no device ({n_devices} visible)")


def aggregate_step_stats(dist, world, device, elapsed_s, pixels_per_step, kernel_ms, verified):
    """What rank 0 reports for a timed region: elapsed = MAX over ranks (the job is as slow as its slowest rank), pixels =
    SUM over ranks (whole-job throughput), verified = AND over ranks, kernel_ms min / max over ranks and every rank's own elapsed time
    (a straggler GPU shows)."""
    import torch
    if world == 1:
        return {"elapsed_s": float(elapsed_s), "pixels_per_step": float(pixels_per_step), "verified": bool(verified),
                "kernel_ms_min": float(kernel_ms), "kernel_ms_max": float(kernel_ms), "kernel_ms_by_rank": [float(kernel_ms)],
                "elapsed_s_by_rank": [float(elapsed_s)]}
    mx = torch.tensor([elapsed_s, kernel_ms, 0.0 if verified else 1.0], dtype=torch.float64, device=device)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    mn = torch.tensor([kernel_ms], dtype=torch.float64, device=device)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    sm = torch.tensor([pixels_per_step], dtype=torch.float64, device=device)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    mine = torch.tensor([kernel_ms, elapsed_s], dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return {"elapsed_s": float(mx[0].item()), "pixels_per_step": float(sm[0].item()), "verified": bool(mx[2].item() == 0.0),
            "kernel_ms_min": float(mn[0].item()), "kernel_ms_max": float(mx[1].item()), "kernel_ms_by_rank": [float(t[0].item()) for t in parts],
            "elapsed_s_by_rank": [float(t[1].item()) for t in parts]}
