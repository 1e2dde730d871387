#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X: Mpixels/s warped, piecewise-affine, 4K RGBA.

A "step" is one pass of the hot path over one batch of synthetic input: F destination point sets of the C3 workload
(3840x2160 RGBA, 11x11-point sinusoidal grid = 200 triangles; frame f uses sin((8 + f mod 4) x / pi): 4 distinct point
sets cycled over the F frames, the pattern of the reference's own harness test/benchmark.js:68,107-110).  Per step and
per frame the library does what the reference redoes per frame (`setDestinyPoints(dst_f); warp()`): per-triangle affine
solves + inverses + triangle spans (k_tri_spans) and the inverse piecewise warp (k_pw_rows, the dominant kernel); for the
projective config C2 the 8x8 DLT solve of every frame (k_solve_frames) is part of the step as well.  Inputs (source
RGBA, meshes, destination points) are resident in HBM before the timed region; outputs stay in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--config C3|C4|C5|C2] [--sources both|shared|distinct]
                    [--points both|resident|fresh]

--gpus N > 1 without a torchrun environment re-executes itself under
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...
(one rank per GPU over RCCL); launched by torchrun directly it reads RANK/LOCAL_RANK/WORLD_SIZE as usual.

Two source layouts are measured:
  shared    BASELINE.json's configuration: all F frames warp ONE source image (test/benchmark.js:107-110).  This is `value`
            and `roofline`.  Its source reads are largely served by L2 / the 256 MiB Infinity Cache, so `roofline` also
            carries `hbm_compulsory_frac` (output written once + source read once).
  distinct  the video case (README.md:121-137): every frame has its own 33 MB source (F x 33 MB >> Infinity Cache), where
            the algorithmic bytes ARE HBM bytes: `roofline_distinct`.
`--points fresh` (part of the default run) times the reference's own loop shape as well: a NEW destination point set goes up
inside every timed step (hg_piecewise_set_frames: host copy into page-locked staging + stream-ordered upload, no GPU wait),
reported as `roofline_fresh` next to the resident-points line (`value`).
After each timed region the bytes the timed kernels wrote are checked (untimed): frame 0 against the reference-generated
golden SHA-256 (tests/golden/golden.json; a data fixture, not the oracle), the other frames against frame f mod 4, the
distinct-source frames through XOR-linearity against the shared-source frames.  `verified` must be true; exit status 3
otherwise.

Multi-GPU: frames shard across ranks (independent units, weak scaling: F frames per GPU); the only exchange is the
one-off broadcast of the shared source texture over RCCL (scatter + all_gather so each xGMI link carries 1/N of it),
done before the timed region and reported as `broadcast_ms`.

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline`, `roofline_distinct` and `cpu_baseline`.
"""
import argparse
import hashlib
import importlib.util
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "homography.js_amd")
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
# reference-generated goldens whose k-th warp is frame k of the config's sequence on rank 0 (tests/golden/gen_golden.mjs)
GOLDEN_CASE = {"C3": "C3_batch_4k", "C2": "C2_projective_1080p", "C5": "C5_batch_8k", "C4": "C4_orbit_4k"}
HBM_ACHIEVABLE_GBS = 6300.0    # what a streaming kernel reaches on this part (same guide); prices `write_floor_ms`
# what `python bench.py` measures besides the headline: the other BASELINE.json configs on one GPU (frames per GPU per step), and -- with
# --gpus N > 1 -- north_star's own sharded lines: C4 batch 512 and C5 batch 64, STRONG scaling, fan-out of the shared source priced in end_to_end
DEFAULT_ALSO = "C4:64,C5:8,C2:64"
DEFAULT_ALSO_STRONG = "C4:512:strong,C5:64:strong"


def _load(name, path):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _pmc_traffic(config, frames, sources):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/hbm_traffic.json:
    FETCH_SIZE / WRITE_SIZE in separate runs of this very command, corrected by the factors tools/calib_fetch measured
    for this library's 4 B/lane access forms).  rocprofv3 cannot run inside the timed process, so this is a number FROM A
    PROFILE of the same workload, labelled as such; null when no matching profile was committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            entries = json.load(f)
        best = None
        for t in entries if isinstance(entries, list) else [entries]:
            if t.get("config") == config and t.get("frames_per_launch") == frames and t.get("sources", "shared") == sources:
                if best is None or t.get("round", 0) >= best.get("round", 0):
                    best = t
        if best:
            return int(best["hbm_bytes_per_launch"]), f"profiles/hbm_traffic.json round {best.get('round')}: {best.get('source', '')}"
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def _golden_shas(config):
    """{frame index: SHA-256 the reference itself produced for that frame of this config's sequence} (possibly empty)."""
    name = GOLDEN_CASE.get(config)
    if not name:
        return {}
    try:
        with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
            for c in json.load(f)["cases"]:
                if c["name"] == name:
                    ids = c.get("orbitFrames") or list(range(len(c["warps"])))
                    return {int(i): w["out"]["sha"] for i, w in zip(ids, c["warps"])}
    except (OSError, ValueError, KeyError):
        pass
    return {}


def launch_command(argv, n):
    """The torchrun command line `bench.py --gpus n` re-executes itself under (127.0.0.1 rendezvous, one rank per GPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a != "--launch-dry-run"]


def also_entries(args, world):
    """[(config, frames or batch, "weak" | "strong")] measured after the headline.  --also given: exactly that list ('none' / '' = nothing).
    Not given: the plain command (no workload-selecting flag) takes DEFAULT_ALSO, plus DEFAULT_ALSO_STRONG when it runs on more than one GPU."""
    also = args.also
    if also is None:
        plain = (args.config == "C3" and args.frames == 64 and args.sources == "both" and args.points == "both" and args.scaling == "weak"
                 and not args.batch and not args.no_verify)
        also = (DEFAULT_ALSO + ("," + DEFAULT_ALSO_STRONG if world > 1 else "")) if plain else ""
    out = []
    for item in [a for a in also.split(",") if a and a != "none"]:
        parts = item.split(":")
        if parts[0] not in ("C2", "C3", "C4", "C5", "C5flat") or len(parts) > 3 or (len(parts) == 3 and parts[2] not in ("strong", "weak")):
            raise SystemExit(f"bench.py: --also entry '{item}' is not CONFIG[:FRAMES[:strong]]")
        out.append((parts[0], int(parts[1]) if len(parts) > 1 and parts[1] else args.frames, parts[2] if len(parts) == 3 else "weak"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=64, help="frames (destination point sets) per GPU per step")
    ap.add_argument("--config", default="C3", choices=["C3", "C4", "C5", "C2", "C5flat"])
    ap.add_argument("--sources", default="both", choices=["both", "shared", "distinct"],
                    help="shared: one source for all frames (BASELINE config, `value`); distinct: one source per frame; both (default)")
    ap.add_argument("--points", default="both", choices=["both", "resident", "fresh"],
                    help="resident: destination points uploaded once before the timed region (`value`); fresh: a new point set goes up "
                         "inside EVERY timed step, like the reference's loop setDestinyPoints(dst_i); warp() (`roofline_fresh`); both (default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the (untimed) output checks")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU-baseline sample (0 = auto, ~10-20 s)")
    ap.add_argument("--also", default=None, help="further measurements in the same invocation, reported under \"also\" with the fields of the headline line: "
                                                 "CONFIG:FRAMES (frames per GPU per step, weak) or CONFIG:BATCH:strong (a fixed batch over all GPUs, shared source, "
                                                 "fan-out priced in end_to_end), comma-separated; 'none' = headline only.  Default (the plain command, no workload flag): "
                                                 f"{DEFAULT_ALSO} and, with --gpus N > 1, {DEFAULT_ALSO_STRONG} (north_star's sharded configs)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --frames frames per GPU per step, per-GPU work fixed as N grows; strong: a FIXED batch of --batch frames "
                         "per step split over the N GPUs (north_star: 'batch = 512 / 64 frames sharded over 8 GPUs'), total work fixed")
    ap.add_argument("--batch", type=int, default=0, help="--scaling strong: frames per step over ALL GPUs (0 = 512 for C4, 64 otherwise)")
    ap.add_argument("--launch-dry-run", action="store_true", help="print the torchrun command --gpus N would re-execute under, and exit")
    args = ap.parse_args()

    # ---------------------------------------------------------------- self-launch (the driver calls `python bench.py --gpus N ...`)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = launch_command(sys.argv[1:], args.gpus)
        if args.launch_dry_run:
            print(json.dumps({"launch": cmd}))
            return 0
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        os.execve(cmd[0], cmd, os.environ)               # never returns
    if args.launch_dry_run:
        print(json.dumps({"launch": None}))
        return 0

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback to measure)")
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    hgdist = _load("hg_dist", os.path.join(PKG, "dist.py"))
    n_devices = torch.cuda.device_count()
    try:
        hgdist.check_launch(world, rank, local_rank, n_devices)      # one rank per GPU of ONE node: a mis-launch is loud
    except RuntimeError as e:
        raise SystemExit(f"bench.py: {e}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)     # nccl == RCCL on ROCm

    hg = _load("hgwarp", os.path.join(PKG, "hgwarp.py"))
    wl = _load("hg_workloads", os.path.join(PKG, "workloads.py"))
    env = dict(torch=torch, dist=dist, world=world, rank=rank, local_rank=local_rank, dev=dev, n_devices=n_devices, hgdist=hgdist, hg=hg, wl=wl)
    line, ok = measure(args, args.config, args.frames, env, primary=True)
    # "also": the other configs of BASELINE.json in the SAME invocation (one lease yields every driver-timed line); each is a complete
    # measurement (own context, own checks) with the fields of the headline.  The plain command takes the default list.
    for name, fr, sc in also_entries(args, world):
        if sc == "strong":
            extra, ok_x = measure(args, name, fr, env, primary=False, scaling="strong", batch=fr, sources="shared", points="resident")
        else:
            extra, ok_x = measure(args, name, fr, env, primary=False, scaling="weak")
        ok = ok and ok_x
        if rank == 0:
            line.setdefault("also", []).append(extra)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if (ok or args.no_verify) else 3


def measure(args, config, F, env, primary=True, scaling=None, batch=None, sources=None, points=None):
    """One complete measurement of `config` with F frames per GPU per step (scaling "strong": a fixed batch over all GPUs): inputs -> HBM,
    timed regions, untimed checks, CPU baseline (headline only).  scaling / batch / sources / points default to the command line's.
    Returns (the JSON line as a dict -- meaningful on rank 0 --, verified)."""
    scaling = scaling or args.scaling
    sources = sources or args.sources
    points = points or args.points
    batch_arg = args.batch if batch is None else batch
    torch, dist, world, rank, local_rank, dev = env["torch"], env["dist"], env["world"], env["rank"], env["local_rank"], env["dev"]
    n_devices, hgdist, hg, wl = env["n_devices"], env["hgdist"], env["hg"], env["wl"]
    cfg = wl.CONFIGS[config]
    W, H = cfg["W"], cfg["H"]
    do_shared, do_distinct = sources in ("both", "shared"), sources in ("both", "distinct")

    # ---------------------------------------------------------------- inputs -> HBM (untimed)
    img_t = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
    if rank == 0:
        img_t.copy_(torch.from_numpy(wl.lcg_image(W, H, 1)))
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    ctx = hg.Context(local_rank, stream=stream.cuda_stream)
    if world == 1:
        ctx.set_image_device(img_t.data_ptr(), W, H)         # (N > 1: attached below, once the broadcast has delivered it)

    piecewise = cfg["kind"] in ("piecewise", "face")
    batch = (batch_arg or (512 if config == "C4" else 64)) if scaling == "strong" else F * world
    frame_ids = hgdist.job_frame_ids(scaling, rank, world, F, batch)     # different ranks get different frames of the same sequence
    F = len(frame_ids)                                       # (strong scaling: block sizes differ by at most one between ranks)
    if F == 0:
        raise SystemExit(f"bench.py: --batch {batch} leaves rank {rank} of {world} without a frame")
    if piecewise:
        if cfg["kind"] == "face":
            sp = wl.face_mesh(W, H, cfg["landmarks"])
            tris = hg.triangulate(sp)                                  # host Delaunay, where the reference calls Delaunator
            seq = wl.face_frames(sp, W, cfg["total_frames"])
            frames = [seq[i % len(seq)] for i in frame_ids]
            same_as = [None] * F
            mesh_txt = f"{cfg['landmarks']}-landmark face mesh"
            pts_txt = f"frames {frame_ids[0]}..{frame_ids[-1]} of the {cfg['total_frames']}-frame orbit"
        else:
            sp, tris = wl.grid_points(W, H, cfg["nx"], cfg["ny"]), wl.grid_triangles(cfg["nx"], cfg["ny"])
            frames = [wl.sin_grid_dst(W, H, cfg["nx"], cfg["ny"], cfg["A"], 8 + (i % 4)) for i in frame_ids]
            same_as = [None if f < 4 else f % 4 for f in range(F)]     # local frame f has the point set of local frame f mod 4
            mesh_txt = f"{cfg['nx']}x{cfg['ny']}-cell sinusoidal grid"
            pts_txt = "4 distinct destination point sets, sin((8 + f mod 4) x / pi), cycled over the frames (test/benchmark.js:68)"
        geoms = [wl.piecewise_geom(d) for d in frames]
        msx, msy = wl.src_min(sp)
        ctx.piecewise_set_mesh(sp, tris, msx, msy)
        offs, total = hg.pack_offsets(geoms)
        ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
        run_resident = ctx.warp_inverse_piecewise_frames_device
        workload = (f"{config}: {W}x{H} RGBA piecewise-affine, {mesh_txt} "
                    f"({tris.size // 3} triangles), " + (f"{batch} frames/step over {world} GPU(s)" if scaling == "strong" else f"{F} frames/GPU/step"))
    else:
        s4 = wl.corners(W, H)
        d4s = [wl.projective_dst(W, H, 0.0125 * (i % 10)) for i in frame_ids]
        same_as = [None if f < 10 else f % 10 for f in range(F)]
        pts_txt = "10 distinct corner sets cycled over the frames (test/benchmark.js:282-283 pattern, animated)"
        # output windows (calculateTransformLimits :1503-1527 on the forward matrix) are host-side scalar work done at
        # setDestinyPoints time; the INVERSE 8x8 solve the reference repeats on every warp (:994) runs on the device inside
        # the timed step (hg_geometric_set_frames_points -> k_solve_frames)
        geoms = [tuple(int(v) for v in hg.transform_limits(1, hg.solve_projective(s4, d4), W, H)) for d4 in d4s]
        mats = [hg.solve_projective(d4, s4) for d4 in d4s]              # host copies: CPU baseline + cross-check only
        offs, total = hg.pack_offsets(geoms)
        ctx.geometric_set_frames_points(1, np.concatenate(d4s), np.tile(s4, F), geoms, offs)     # inverse: dst -> src (:994)
        solve_txt = ", 8x8 DLT solve per frame on the device inside the step"
        run_resident = ctx.warp_inverse_geometric_frames_device
        workload = (f"{config}: {W}x{H} RGBA projective, 4 corner points, " +
                    (f"{batch} frames/step over {world} GPU(s)" if scaling == "strong" else f"{F} frames/GPU/step") + solve_txt)

    run = run_resident
    # --points fresh: a ring of R point sets (the config's own sequence shifted by k frames), one uploaded per timed step
    fresh_sets = None
    if piecewise and points in ("both", "fresh") and sources != "distinct":
        R = 8
        fresh_sets = []
        for k in range(R):
            if cfg["kind"] == "face":
                fr_k = [seq[(i + k) % len(seq)] for i in frame_ids]
            else:
                fr_k = [wl.sin_grid_dst(W, H, cfg["nx"], cfg["ny"], cfg["A"], 8 + ((i + k) % 4)) for i in frame_ids]
            g_k = [wl.piecewise_geom(d) for d in fr_k]
            o_k, t_k = hg.pack_offsets(g_k)
            fresh_sets.append({"frames": fr_k, "geoms": g_k, "offs": o_k, "total": t_k, "args": ctx.frame_set_args(np.concatenate(fr_k), g_k, o_k)})
        total = max([total] + [fs["total"] for fs in fresh_sets])
    elif not piecewise and points in ("both", "fresh") and sources != "distinct":
        fresh_sets = []                                      # projective: the corner sets of the sequence shifted by k frames
        for k in range(8):
            d_k = [wl.projective_dst(W, H, 0.0125 * ((i + k) % 10)) for i in frame_ids]
            g_k = [tuple(int(v) for v in hg.transform_limits(1, hg.solve_projective(s4, d4), W, H)) for d4 in d_k]
            o_k, t_k = hg.pack_offsets(g_k)
            fresh_sets.append({"geoms": g_k, "offs": o_k, "total": t_k, "args": ctx.geometric_points_args(1, np.concatenate(d_k), np.tile(s4, F), g_k, o_k)})
        total = max([total] + [fs["total"] for fs in fresh_sets])
    out_t = torch.empty(total, dtype=torch.uint8, device=dev)
    d_out = out_t.data_ptr()

    # ---------------------------------------------------------------- N > 1: the one exchange of the job, untimed
    # Shared source texture, rank 0 -> everybody: scatter of 1/N slices + all-gather, so that every xGMI link carries 1/N of
    # the image.  The root does not wait for it: it already holds the image, so its first step is queued on the warp stream
    # BEFORE the collectives start on torch's stream and runs under them (`broadcast_hidden_ms` = how much of the two
    # overlapped); the other ranks attach the texture when it has arrived.
    broadcast = None
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t0 = time.perf_counter()
        if rank == 0:
            ctx.set_image_device(img_t.data_ptr(), W, H)
            ev[0].record(stream)
            run(d_out)                                       # the root's first frames, under the fan-out
            ev[1].record(stream)
        ev[2].record()
        img_full = hgdist.broadcast_source(img_t, rank, world, dist, verify=False)
        ev[3].record()
        torch.cuda.synchronize()
        span_ms = (time.perf_counter() - t0) * 1e3
        b_ms = ev[2].elapsed_time(ev[3])
        s_ms = ev[0].elapsed_time(ev[1]) if rank == 0 else 0.0
        ctx.sync()
        img_t = img_full
        hgdist.verify_replicas(img_t, dist)                 # untimed: every rank holds the same bytes, or this raises
        ctx.set_image_device(img_t.data_ptr(), W, H)
        bt = torch.tensor([b_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(bt, op=dist.ReduceOp.MAX)
        broadcast = {"broadcast_ms": round(float(bt.item()), 3), "root_first_step_ms": round(s_ms, 3), "root_span_ms": round(span_ms, 3),
                     "broadcast_hidden_ms": round(max(0.0, min(b_ms, s_ms, b_ms + s_ms - span_ms)), 3) if rank == 0 else None,
                     "image_bytes": int(W * H * 4)}
    broadcast_ms = broadcast["broadcast_ms"] if broadcast else 0.0
    n_out = [g[2] * g[3] for g in geoms]
    px_per_step = float(sum(n_out))

    def frame_view(t, f):
        return t[offs[f]: offs[f] + n_out[f] * 4]

    # N_hit per frame (algorithmic read bytes): alpha==255 count when the same frames run on an all-255 source (untimed)
    solid = torch.full((H, W, 4), 255, dtype=torch.uint8, device=dev)
    ctx.set_image_device(solid.data_ptr(), W, H)
    run(d_out)
    ctx.sync()
    hit_masks = [frame_view(out_t, f).view(-1, 4)[:, 3] == 255 for f in range(F)]
    n_hit = [int(m.sum().item()) for m in hit_masks]
    if not do_distinct or args.no_verify:
        hit_masks = None
    ctx.set_image_device(img_t.data_ptr(), W, H)
    del solid
    algo_bytes_per_launch = float(sum(4 * no + 4 * nh for no, nh in zip(n_out, n_hit)))
    compulsory_bytes_shared = float(sum(4 * no for no in n_out)) + 4.0 * W * H      # every output byte once + the shared source once

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(run=None):
        """ramp -> W warmup steps -> K timed steps (barrier + synchronize on both sides); returns (elapsed_s, kernel_ms, launches)"""
        run = run or run_resident
        # The GPU sat idle during the untimed host work: ~50 ms of the same steps first, so that warmup and the timed region
        # run at the sustained clocks (a 0.6 ms step is far shorter than the power-state ramp).  Untimed.
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < 0.05:
            for _ in range(16):
                run(d_out)
            ctx.sync()
        for _ in range(args.warmup):
            run(d_out)
        ctx.sync()
        ctx.set_timing(True)                       # hipEvent pairs around the dominant kernel, on the launch stream
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run(d_out)
        ctx.sync()                                 # waits for the stream (and settles any frame the fused path deferred)
        barrier()
        elapsed = time.perf_counter() - t0
        k_total_ms, k_launches = ctx.kernel_ms_stats()
        ctx.set_timing(False)
        k_ms = k_total_ms / max(k_launches, 1)
        # MAX of the elapsed time over ranks, SUM of the pixels, min / max of the kernel time (homography.js_amd/dist.py; the same
        # function runs over gloo in tests/test_dist_cpu.py)
        agg = hgdist.aggregate_step_stats(dist, world, dev, elapsed, px_per_step, k_ms, True)
        region_stats.append(agg)
        return agg["elapsed_s"], k_ms, k_launches

    kernel_name = None
    region_stats = []

    def step_frac(elapsed_s):
        """The whole step against the roofline, not only its dominant kernel: algorithmic bytes of one step / wall time per step
        (producer kernel, kernel boundaries and host launch gaps included) / peak.  This is the fraction `value` corresponds to."""
        return round(algo_bytes_per_launch / (elapsed_s / args.steps) / 1e9 / HBM_PEAK_GBS, 4) if elapsed_s > 0 else 0.0

    def roofline_block(k_ms, launches, sources, elapsed_s):
        achieved = algo_bytes_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic, traffic_src = _pmc_traffic(config, F, sources)
        # fabric_frac: the bytes that really crossed the memory fabric (PMC pass of the same workload, incl. Infinity-Cache hits) per launch /
        # THIS run's kernel time / peak -- a fraction that cannot exceed 1, next to the algorithmic `frac` that can when frames share a source;
        # write_floor_ms: the output alone (4 * sum N_out, written exactly once) at the ~6.3 TB/s a streaming kernel reaches on this part
        write_bytes = float(sum(4 * no for no in n_out))
        return {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "step_frac": step_frac(elapsed_s),
                "step_ms": round(elapsed_s * 1e3 / args.steps, 4), "traffic": traffic, "traffic_source": traffic_src,
                "fabric_frac": round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic and k_ms > 0 else None,
                "write_floor_ms": round(write_bytes / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3, 5),
                "kernel_ms_over_write_floor": round(k_ms / (write_bytes / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3), 3) if write_bytes > 0 else None,
                "kernel_ms": round(k_ms, 5), "launches_timed": launches, "algorithmic_bytes_per_launch": int(algo_bytes_per_launch),
                "note": "achieved = (4*N_out + 4*N_hit summed over the frames of one launch) / mean hipEvent duration of that kernel; "
                        "step_frac = the same bytes / wall time per step (all kernels of the step, boundaries, launch gaps) / peak; "
                        "fabric_frac = `traffic` (measured fabric bytes per launch, from the committed PMC pass) / kernel_ms / peak (<= 1 by construction); "
                        "write_floor_ms = 4 * sum N_out / 6.3 TB/s (the output stream alone at the achievable rate)"}

    verified, checks = True, []

    def check(name, ok):
        nonlocal verified
        checks.append({"check": name, "ok": bool(ok)})
        verified = verified and bool(ok)

    # ---------------------------------------------------------------- shared source: BASELINE.json's configuration -> `value`
    px_all = px_per_step
    if world > 1:
        px = torch.tensor([px_per_step], dtype=torch.float64, device=dev)
        dist.all_reduce(px, op=dist.ReduceOp.SUM)
        px_all = float(px.item())
    res = {}
    shared_copy = None
    if do_shared:
        elapsed, k_ms, k_launches = timed_region()
        kernel_name = ({3: "k_pw_patch", 4: "k_pw_fused", 5: "k_pw_tile"}.get(ctx.last_piecewise_kernel(), "k_pw_rows") if piecewise else "k_geo_fast<projective>")
        rf = roofline_block(k_ms, k_launches, "shared", elapsed)
        comp = compulsory_bytes_shared / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        rf["hbm_compulsory_frac"] = round(comp / HBM_PEAK_GBS, 4)
        rf["note"] += ("; all frames share ONE source, whose reads are largely served by L2 / the 256 MiB Infinity Cache, so the ALGORITHMIC "
                       "rate (SURVEY.md 8d: the source read is counted per frame) can exceed the HBM peak -- frac > 1 then says 'faster than a "
                       "kernel that fetched every source pixel from HBM could be', not 'HBM at more than 100 %': hbm_compulsory_frac prices only "
                       "what must cross HBM (every output byte once + the source once) and `traffic` is what did (PMC); "
                       "roofline_distinct is the layout where algorithmic bytes are HBM bytes")
        res["shared"] = (elapsed, rf)
        if not args.no_verify:                      # the bytes the timed kernels wrote (untimed checks)
            want = {i: h for i, h in _golden_shas(config).items() if i in frame_ids}
            if want:
                ok = all(hashlib.sha256(frame_view(out_t, frame_ids.index(i)).cpu().numpy().tobytes()).hexdigest() == h for i, h in want.items())
                check(f"frames {sorted(want)} of the timed output == reference goldens {GOLDEN_CASE[config]} (sha256)", ok)
            for f in range(F):
                if same_as[f] is not None and n_out[f] == n_out[same_as[f]]:
                    if not torch.equal(frame_view(out_t, f), frame_view(out_t, same_as[f])):
                        check(f"frame {f} == frame {same_as[f]} (same point set)", False)
                        break
            else:
                check("every frame == the frame with its point set among the first few (f mod period)", True)
            if do_distinct or fresh_sets:
                shared_copy = out_t.clone()

    # ---------------------------------------------------------------- fresh destination points inside every timed step
    # What the reference's loop does (test/benchmark.js:107-110: setDestinyPoints(dst_i); warp()): per step one
    # hg_piecewise_set_frames (host copy into page-locked staging + stream-ordered upload, no GPU wait) and the two kernels.
    fresh = None
    if do_shared and fresh_sets:
        step_i = [0]

        def run_fresh(d):
            fs = fresh_sets[step_i[0] % len(fresh_sets)]
            step_i[0] += 1
            if piecewise:
                ctx.piecewise_set_frames_prepared(fs["args"])
                ctx.warp_inverse_piecewise_frames_device(d)
            else:
                ctx.geometric_set_frames_points_prepared(fs["args"])
                ctx.warp_inverse_geometric_frames_device(d)
        walks0, redone0 = ctx.layout_walks(), ctx.redone_frames()
        n_pts_txt = f"{sp.size // 2} points" if piecewise else "2 x 4 corner points"
        elapsed_f, k_ms_f, k_launches_f = timed_region(run_fresh)
        k_last = (step_i[0] - 1) % len(fresh_sets)
        fs = fresh_sets[k_last]
        fresh = {"ms_per_step": round(elapsed_f * 1e3 / args.steps, 4), "kernel_ms": round(k_ms_f, 5), "step_frac": step_frac(elapsed_f),
                 "value_mpixels_per_s": round(px_all * args.steps / elapsed_f / 1e6, 1),
                 "vs_resident_ms_per_step": round(elapsed_f / res["shared"][0], 4),
                 "layout_walks_in_region": ctx.layout_walks() - walks0, "frames_redone_in_region": ctx.redone_frames() - redone0,
                 "point_sets": f"ring of {len(fresh_sets)} sets (the config's sequence shifted by k frames), one hg_{'piecewise_set_frames' if piecewise else 'geometric_set_frames_points'} per step: "
                               f"{F} frames x {n_pts_txt} copied to page-locked staging and uploaded stream-ordered, no GPU wait"}
        if not args.no_verify and shared_copy is not None:
            ok, n_cmp = True, 0
            for f in range(F):
                fp = f + k_last if cfg["kind"] == "face" else ((f + k_last) % (4 if piecewise else 10))     # resident frame with the same point set
                if fp >= F or fs["geoms"][f] != tuple(geoms[fp]):
                    continue
                nb = n_out[fp] * 4
                n_cmp += 1
                if not torch.equal(out_t[fs["offs"][f]: fs["offs"][f] + nb], shared_copy[offs[fp]: offs[fp] + nb]):
                    ok = False
                    break
            if n_cmp > 0:
                check(f"fresh-points step (set {k_last}): {n_cmp} frames == the resident-points frames with the same point set", ok)
        if piecewise:                                        # back to the resident set
            ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
        else:
            ctx.geometric_set_frames_points(1, np.concatenate(d4s), np.tile(s4, F), geoms, offs)

    # ---------------------------------------------------------------- distinct sources: one 4*W*H source per frame (video case)
    if do_distinct:
        srcs = img_t.unsqueeze(0).repeat(F, 1, 1, 1)                    # F x H x W x 4; image f = image 0 XOR c_f (c_0 = 0)
        consts = [0] + [((f * 37 + 11) & 0xFF) or 1 for f in range(1, F)]
        for f in range(1, F):
            srcs[f] ^= consts[f]
        ctx.set_images_device(srcs.data_ptr(), W, H, F, W * H * 4)
        elapsed_d, k_ms_d, k_launches_d = timed_region()
        kernel_shared = kernel_name                 # (the layout policy may pick another kernel when every frame streams its own source)
        kernel_name = ({3: "k_pw_patch", 4: "k_pw_fused", 5: "k_pw_tile"}.get(ctx.last_piecewise_kernel(), "k_pw_rows") if piecewise else "k_geo_fast<projective>")
        rd = roofline_block(k_ms_d, k_launches_d, "distinct", elapsed_d)
        kernel_name = kernel_shared or kernel_name
        rd["sources"] = f"{F} distinct {W}x{H} RGBA sources per step ({F * W * H * 4 / 1e6:.0f} MB >> 256 MiB Infinity Cache): algorithmic bytes are HBM bytes"
        rd["ms_per_step"] = round(elapsed_d * 1e3 / args.steps, 4)
        rd["value_mpixels_per_s"] = round(px_all * args.steps / elapsed_d / 1e6, 1)
        res["distinct"] = (elapsed_d, rd)
        if not args.no_verify and hit_masks is not None:
            # XOR-linearity of a pure gather: out(img ^ c) == out(img) ^ (c on pixels that read a source pixel, 0 elsewhere)
            base = shared_copy
            if base is None:                        # --sources distinct: produce the shared-source frames now (untimed)
                ctx.set_image_device(img_t.data_ptr(), W, H)
                base = torch.empty_like(out_t)
                run(base.data_ptr())
                ctx.sync()
            ok = True
            for f in range(F):
                d = (frame_view(out_t, f) ^ frame_view(base, f)).view(-1, 4)
                exp = (hit_masks[f].to(torch.uint8) * consts[f]).unsqueeze(1).expand(-1, 4)
                if not torch.equal(d, exp):
                    ok = False
                    break
            check("distinct-source frames == shared-source frames XOR per-frame constant on exactly the hit pixels (all frames)", ok)
            # ... and against the reference's own goldens: frame f of the TIMED output with its constant XOR-ed back on the hit
            # pixels is what the un-XOR-ed source gives (the goldens were generated from that source)
            want = {i: h for i, h in _golden_shas(config).items() if i in frame_ids}
            if want:
                ok = True
                for i, h in want.items():
                    f = frame_ids.index(i)
                    undo = (hit_masks[f].to(torch.uint8) * consts[f]).unsqueeze(1).expand(-1, 4)
                    rec = frame_view(out_t, f).view(-1, 4) ^ undo
                    ok = ok and hashlib.sha256(rec.contiguous().cpu().numpy().tobytes()).hexdigest() == h
                check(f"frames {sorted(want)} of the timed one-source-per-frame output, constants XOR-ed back on the hit pixels, == reference goldens {GOLDEN_CASE[config]} (sha256)", ok)
        del srcs, shared_copy
        ctx.set_image_device(img_t.data_ptr(), W, H)

    primary_src = "shared" if "shared" in res else "distinct"
    elapsed, roofline = res[primary_src]
    ms_per_step = elapsed * 1e3 / args.steps
    value = px_all * args.steps / elapsed / 1e6          # Mpixels/s, whole job
    if world > 1:
        v = torch.tensor([1.0 if verified else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        verified = bool(v.item() == 1.0)

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1 only): the oracle, 1 core, bounded sample
    cpu = None
    if primary and rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from hgtest import oracle as O             # checker / baseline only; never on the measured GPU path
        img = img_t.cpu().numpy()
        t_budget, n_done, px_done = 12.0, 0, 0
        t0 = time.perf_counter()
        max_frames = args.cpu_frames or 10 ** 9
        while n_done < max_frames and (time.perf_counter() - t0 < t_budget or n_done == 0):
            f = n_done % F
            if piecewise:
                O.warp_inverse_piecewise(sp, frames[f], tris, img, msx, msy, *geoms[f])
            else:
                O.warp_inverse_geometric(1, mats[f], img, *geoms[f])
            px_done += n_out[f]
            n_done += 1
        dt = time.perf_counter() - t0
        cpu = {"value": round(px_done / dt / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
               "sample": f"{n_done} frames of the same workload through oracle/hg_oracle.c (C restatement of the reference's JS loops, "
                         f"gcc -O2, single thread) in {dt:.1f} s; the reference itself is single-threaded JavaScript "
                         f"(24.3 Mpix/s on C3 under Node 12, BASELINE.md §2)"}
        # the same C restatement on every host core: one frame per thread at a time (ctypes releases the GIL), ~6 s sample
        if piecewise:
            from concurrent.futures import ThreadPoolExecutor
            cores = os.cpu_count() or 1
            deadline = time.perf_counter() + 6.0

            def _work(k):
                done, j = 0, 0
                while j == 0 or time.perf_counter() < deadline:
                    f = (k + j) % F
                    O.warp_inverse_piecewise(sp, frames[f], tris, img, msx, msy, *geoms[f])
                    done += n_out[f]
                    j += 1
                return done, j
            t1 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                r_mt = list(ex.map(_work, range(cores)))
            dt_mt = time.perf_counter() - t1
            cpu["all_cores"] = {"value": round(sum(r[0] for r in r_mt) / dt_mt / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port",
                                "sample": f"{sum(r[1] for r in r_mt)} frames on {cores} threads (one frame per thread at a time, "
                                          f"fresh output buffers per frame) in {dt_mt:.1f} s"}
        # the same algorithm as plain JavaScript under this box's Node (what the reference's own loops achieve here)
        if piecewise and config in ("C3", "C5"):
            import shutil
            import subprocess
            node = shutil.which("node")
            if node:
                try:
                    p = subprocess.run([node, os.path.join(ROOT, "oracle", "hg_oracle_js.mjs"), "bench", config, "8"],
                                       capture_output=True, text=True, timeout=120)
                    j = json.loads(p.stdout.strip().splitlines()[-1])
                    cpu["node"] = {"value": j["mpix_per_s"], "unit": "Mpixels/s", "cores": 1, "node": j["node"],
                                   "sample": f"{j['frames']} frames in {j['seconds']} s through oracle/hg_oracle_js.mjs (JS restatement, pinned to the goldens)"}
                except Exception as e:                      # noqa: BLE001 - the baseline is informative only
                    cpu["node"] = {"error": str(e)[:200]}

    step_ms_by_rank = [round(e * 1e3 / args.steps, 5) for e in region_stats[0]["elapsed_s_by_rank"]] if region_stats else None
    # who ran what: every rank's device and block of frames (a heterogeneous or partitioned node, or a mis-sharded batch, shows here)
    props = torch.cuda.get_device_properties(dev)
    ranks = hgdist.gather_rank_info(dist, world, {"rank": rank, "device": local_rank, "device_name": props.name, "xcc": ctx.xcc_count(),
                                                 "cus": props.multi_processor_count, "frames": F, "first_frame": frame_ids[0], "pixels_per_step": int(px_per_step)})
    e2e_ms, e2e_value = hgdist.end_to_end(ms_per_step, broadcast_ms, px_all)
    line = {"metric": "Mpixels/s warped (piecewise-affine, 4K RGBA)" if config == "C3" else f"Mpixels/s warped ({config})",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u8 pixels / f64 coordinates", "data": "synthetic",
            "config": {"workload": workload + (" on a shared source" if primary_src == "shared" else ", one source per frame"),
                       "frames_per_gpu_per_step": F if scaling == "weak" else None, "frames_per_step_all_gpus": batch, "point_sets": pts_txt,
                       "value_is": ("WEAK scaling: every GPU warps its own %d frames per step, so `value` grows ~N x by construction; the one-off fan-out of the shared source is "
                                    "EXCLUDED (inputs resident) and priced in end_to_end; north_star's fixed-batch lines are the \"scaling\": \"strong\" entries under `also`" % F)
                                   if scaling == "weak" else
                                   ("STRONG scaling: a fixed batch of %d frames per step split over the GPUs; `value` excludes the one-off fan-out of the shared source, "
                                    "end_to_end includes it" % batch),
                       "rccl_world": world, "ranks": ranks,
                       "source_fanout": None if world == 1 else f"rank 0 -> {world} ranks: dist.scatter of 1/{world} slices + all_gather_into_tensor over RCCL (every xGMI link carries 1/{world} of the image)",
                       "end_to_end": {"ms_per_batch_incl_broadcast": round(e2e_ms, 4), "value_mpixels_per_s_incl_broadcast": round(e2e_value, 1),
                                      "note": "one step + the one-off fan-out of the shared source (broadcast_ms), as if every batch shipped a new source texture; `value` excludes it (inputs resident)"},
                       "output_pixels_per_step_per_gpu": int(px_per_step), "sources": primary_src,
                       "parallelism": f"frames sharded over {world} GPU(s); shared source broadcast once (scatter+all_gather over RCCL)",
                       "broadcast_ms": round(broadcast_ms, 3), "broadcast": broadcast, "gpus_on_node": n_devices,
                       "ms_per_step_by_rank": step_ms_by_rank,
                       "kernel_ms_over_ranks": {"min": round(region_stats[0]["kernel_ms_min"], 5), "max": round(region_stats[0]["kernel_ms_max"], 5),
                                                "by_rank": [round(v, 5) for v in region_stats[0]["kernel_ms_by_rank"]]} if region_stats else None},
            "verified": None if args.no_verify else verified, "checks": checks,
            "roofline": roofline, "roofline_fresh": fresh, "roofline_distinct": res["distinct"][1] if "distinct" in res and primary_src != "distinct" else None,
            "cpu_baseline": cpu}
    ctx.close()
    return line, bool(verified or args.no_verify)


if __name__ == "__main__":
    sys.exit(main())
