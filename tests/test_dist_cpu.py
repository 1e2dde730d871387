"""N > 1 path on CPU (gloo, world_size 2): frame sharding covers every frame exactly once and the scatter + all-gather
source broadcast delivers the root's texture bit-for-bit to every rank."""
import importlib.util
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "homography.js_amd", rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def test_shard_frames_partition():
    D = _load("hg_dist", "dist.py")
    for n in (0, 1, 7, 8, 64, 512, 513):
        for world in (1, 2, 3, 8):
            owned = [list(D.shard_frames(n, r, world)) for r in range(world)]
            flat = [f for o in owned for f in o]
            assert flat == list(range(n))
            assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1


def _worker(rank, world, port, shape, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = _load("hg_dist", "dist.py")
        WL = _load("hg_workloads", "workloads.py")
        h, w = shape
        want = torch.from_numpy(WL.lcg_image(w, h, 5))
        img = want.clone() if rank == 0 else torch.zeros_like(want)
        got = D.broadcast_source(img, rank, world, dist, verify=True)
        ok = bool(torch.equal(got, want))
        bad = got.clone()
        if rank == 1:
            bad.view(-1)[5] ^= 1                               # one flipped bit on one rank must be noticed by every rank
        try:
            D.verify_replicas(bad, dist)
            ok = False
        except RuntimeError:
            pass
        frames = list(D.shard_frames(11, rank, world))
        q.put((rank, ok, frames))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape", [(37, 53), (64, 64)])       # a size that does not divide by the world size, and one that does
def test_broadcast_source_gloo_world2(shape):
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sorted(f for _, _, fr in res for f in fr) == list(range(11))


def _root_worker(rank, world, port, shape, src, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = _load("hg_dist", "dist.py")
        WL = _load("hg_workloads", "workloads.py")
        h, w = shape
        want = torch.from_numpy(WL.lcg_image(w, h, 9))
        img = want.clone() if rank == src else torch.full_like(want, 0xAB)      # only the root holds the texture; the others hold garbage
        got = D.broadcast_source(img, rank, world, dist, src=src, verify=True)
        agg = D.aggregate_step_stats(dist, world, torch.device("cpu"), elapsed_s=0.125 * (rank + 1), pixels_per_step=100.0, kernel_ms=1.0 + rank, verified=True)
        q.put((rank, bool(torch.equal(got, want)), list(D.shard_frames(13, rank, world)), agg["elapsed_s_by_rank"], agg["kernel_ms_by_rank"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape,src", [((37, 53), 2), ((64, 64), 3)])     # ragged and exact slice sizes, roots other than rank 0
def test_broadcast_source_gloo_world4_nonzero_root(shape, src):
    """The source fan-out (scatter of 1/N slices + all-gather) from a root that is NOT rank 0, world size 4: every rank ends with the
    root's bytes, whatever it held before; the per-rank bookkeeping (elapsed / kernel ms by rank) arrives in rank order."""
    world = 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_root_worker, args=(r, world, port, shape, src, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _, _ in res)
    assert sorted(f for _, _, fr, _, _ in res for f in fr) == list(range(13))
    for _, _, _, el, km in res:
        assert el == [0.125, 0.25, 0.375, 0.5] and km == [1.0, 2.0, 3.0, 4.0]


def _bookkeeping_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = _load("hg_dist", "dist.py")
        F = 5
        ids = D.rank_frame_ids(rank, F)
        # every rank reports its own numbers; rank 1 is the straggler and the one whose output check failed
        agg = D.aggregate_step_stats(dist, world, torch.device("cpu"), elapsed_s=0.25 + 0.5 * rank, pixels_per_step=1000.0 * (rank + 1),
                                     kernel_ms=0.5 + 0.125 * rank, verified=(rank == 0))
        ok_all = D.aggregate_step_stats(dist, world, torch.device("cpu"), 0.5, 10.0, 1.0, True)
        q.put((rank, ids, agg, ok_all["verified"]))
    finally:
        dist.destroy_process_group()


def test_bench_bookkeeping_gloo_world2():
    """bench.py's N > 1 bookkeeping on CPU tensors over gloo: frame ids per rank (weak scaling: disjoint blocks of the sequence),
    elapsed = MAX over ranks, pixels = SUM, verified = AND, kernel ms min / max / by rank -- identical on every rank."""
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bookkeeping_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2, 3, 4] and res[1][1] == [5, 6, 7, 8, 9]
    for _, _, agg, ok_all in res:
        assert agg["elapsed_s"] == 0.75 and agg["pixels_per_step"] == 3000.0 and agg["verified"] is False
        assert agg["kernel_ms_min"] == 0.5 and agg["kernel_ms_max"] == 0.625 and agg["kernel_ms_by_rank"] == [0.5, 0.625]
        assert ok_all is True


def _strong_worker(rank, world, port, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = _load("hg_dist", "dist.py")
        ids = D.job_frame_ids("strong", rank, world, 64, batch)
        # a rank's step takes as long as it has frames; its pixels are 10 per frame
        agg = D.aggregate_step_stats(dist, world, torch.device("cpu"), elapsed_s=0.01 * len(ids), pixels_per_step=10.0 * len(ids), kernel_ms=0.1 * len(ids), verified=True)
        info = D.gather_rank_info(dist, world, {"rank": rank, "frames": len(ids), "first_frame": ids[0] if ids else None})
        q.put((rank, ids, agg, info))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 7), (2, 64), (4, 10)])
def test_strong_scaling_bookkeeping_gloo(world, batch):
    """bench.py --scaling strong: a FIXED batch split over the ranks (north_star: "batch = 512 / 64 frames sharded over 8 GPUs") -- every
    frame owned exactly once, block sizes differ by at most one, pixels SUM to the whole batch, elapsed = the largest block's time,
    every rank sees every rank's block (gather_rank_info); and the end-to-end figure adds the source fan-out once per batch."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [f for _, ids, _, _ in res for f in ids] == list(range(batch))
    sizes = [len(ids) for _, ids, _, _ in res]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == batch
    for _, _, agg, info in res:
        assert agg["pixels_per_step"] == 10.0 * batch and abs(agg["elapsed_s"] - 0.01 * max(sizes)) < 1e-12
        assert [i["rank"] for i in info] == list(range(world)) and [i["frames"] for i in info] == sizes
    D = _load("hg_dist", "dist.py")
    assert D.job_frame_ids("weak", 1, 2, 5, 999) == [5, 6, 7, 8, 9]
    ms, v = D.end_to_end(2.0, 0.5, 5.0e6)
    assert ms == 2.5 and abs(v - 2000.0) < 1e-9


def test_check_launch_is_loud():
    D = _load("hg_dist", "dist.py")
    D.check_launch(1, 0, 0, 1)
    D.check_launch(2, 1, 1, 8)                      # 2 of the 8 GPUs of a node: fine
    D.check_launch(8, 7, 7, 8)
    for bad in ((8, 0, 0, 4), (2, 2, 0, 8), (2, 1, 5, 4), (1, 0, -1, 1)):
        with pytest.raises(RuntimeError):
            D.check_launch(*bad)
    # world 1 needs no process group at all
    agg = D.aggregate_step_stats(None, 1, None, 0.5, 7.0, 0.25, True)
    assert agg["elapsed_s"] == 0.5 and agg["kernel_ms_by_rank"] == [0.25] and agg["verified"] is True


def test_bench_self_launch_command():
    """`python bench.py --gpus N` (what the driver runs) re-executes itself under torch.distributed.run: the dry run prints
    the exact command without needing a GPU."""
    import json
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2", "--launch-dry-run"],
                       capture_output=True, text=True, timeout=120, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert p.returncode == 0, p.stderr
    cmd = json.loads(p.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    # under torchrun's environment the same invocation does not re-launch
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launch-dry-run"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, WORLD_SIZE="8", RANK="0", LOCAL_RANK="0"))
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["launch"] is None


def test_bench_default_also_entries():
    """What the driver's one command records: `python bench.py [--gpus N --steps K --warmup W]` measures the headline AND the other
    BASELINE.json configs; with N > 1 also north_star's fixed-batch (strong) lines.  Any workload-selecting flag means "exactly this"."""
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("hg_bench", os.path.join(ROOT, "bench.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)

    def ns(**kw):
        base = dict(config="C3", frames=64, sources="both", points="both", scaling="weak", batch=0, no_verify=False, also=None)
        base.update(kw)
        return argparse.Namespace(**base)
    assert B.also_entries(ns(), 1) == [("C4", 64, "weak"), ("C5", 8, "weak"), ("C2", 64, "weak")]
    assert B.also_entries(ns(), 8) == [("C4", 64, "weak"), ("C5", 8, "weak"), ("C2", 64, "weak"), ("C4", 512, "strong"), ("C5", 64, "strong")]
    for flag in (dict(config="C4"), dict(frames=8), dict(sources="shared"), dict(points="resident"), dict(scaling="strong"), dict(batch=64), dict(no_verify=True)):
        assert B.also_entries(ns(**flag), 8) == [], flag
    assert B.also_entries(ns(also="none"), 8) == [] and B.also_entries(ns(also=""), 1) == []
    assert B.also_entries(ns(also="C4:32,C5:64:strong", frames=16), 2) == [("C4", 32, "weak"), ("C5", 64, "strong")]
    assert B.also_entries(ns(also="C2", frames=16), 1) == [("C2", 16, "weak")]
    with pytest.raises(SystemExit):
        B.also_entries(ns(also="C9:4"), 1)
