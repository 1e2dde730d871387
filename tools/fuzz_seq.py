#!/usr/bin/env python3
"""Sequence fuzz of the queueing logic (run on the GPU box): tools/fuzz_seq.py [sequences] [seed].
Random SEQUENCES of calls on one context -- inverse piecewise batches (fresh point sets, 1..3 frames), forward piecewise batches,
geometric batches, option changes, hg_sync at random points or not at all until the end -- into a small pool of output buffers
that get reused, on meshes that make the device flag frames (folded / thin-triangle meshes for the inverse fast path, a mesh denser
than the forward tile lists).  After every sync, and at the end, every buffer must hold exactly what the CPU oracle computes for the
LAST call that wrote each of its bytes: deferred redos, staged frame sets, the status rings and 'a later call reused this buffer' all
have to be right for that."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hgtest import golden as G, hip, oracle as O, workloads as WL  # noqa: E402

HG = hip.load()
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
calls = 0
redone = 0


def geom_of(d):
    m = O.minmax_xy(d)
    return (int(m[0]), int(m[1]), int(m[2] - m[0]), int(m[3] - m[1]))


for seq in range(n_seq):
    kind = seq % 4
    if kind == 3:                                              # forward tile lists overflow (capacity 64): flagged + redone, capacity grows
        W = H = 256
        nx = ny = 48
    else:
        W, H = int(rng.integers(60, 420)), int(rng.integers(50, 300))
        nx, ny = int(rng.integers(1, 10)), int(rng.integers(1, 8))
    if kind == 1:                                              # 1100 thin triangles crossing every row: more spans per row than the fast path's lists ever hold
        nt, W, H = 1100, 2400, 8
        xs = np.linspace(0, W, nt + 1)
        sp = np.stack([np.repeat(xs, 2), np.tile([0.0, float(H)], nt + 1)], 1)
        tris = np.array([[2 * i, 2 * i + 2, 2 * i + 1] for i in range(nt)], np.uint32).ravel()
        nx, ny = nt, 1
    else:
        sp = WL.grid_points(W, H, nx, ny).reshape(-1, 2).astype(np.float64)
        tris = WL.grid_triangles(nx, ny)
    img = G.lcg_image(W, H, 100 + seq)
    sp32 = sp.astype(np.float32).ravel()
    ms = O.minmax_xy(sp32)
    fmap_w, fmap_h = int(ms[2] - ms[0]), int(ms[3] - ms[1])
    fmap = O.build_tri_map(sp32, tris, fmap_w, int(ms[1]), fmap_w * fmap_h) if fmap_w > 0 and fmap_h > 0 else None

    def new_points():
        jit = rng.uniform(0, 0.4)
        dp = (sp + rng.uniform(-jit, jit, sp.shape) * [W / nx, H / ny]) * rng.uniform(0.5, 1.6, 2) + rng.uniform(-20, 40, 2)
        if kind == 1:
            dp = sp * [rng.uniform(0.9, 1.1), rng.uniform(1.0, 2.5)] + rng.uniform(-3, 3, 2)
        if kind == 2 and rng.random() < 0.5:
            dp[rng.integers(0, dp.shape[0])] += rng.uniform(-60, 60, 2)           # a fold
        return dp.astype(np.float32).ravel()

    c = HG.Context(0)
    try:
        c.set_image(img)
        imgs, d_srcs = [img], 0
        c.piecewise_set_mesh(sp32, tris, int(ms[0]), int(ms[1]))
        if kind == 3:
            c.set_option("fwd_tiles", 1)
        cap = 1 << 21
        bufs = [c.alloc(cap) for _ in range(3)]
        for b in bufs:
            c.to_device(b, np.zeros(cap, np.uint8))
        want = [np.zeros(cap, np.uint8) for _ in range(3)]
        n_ops = int(rng.integers(3, 12))
        for op in range(n_ops + 1):
            last = op == n_ops
            r = rng.random()
            if last or r < 0.2:
                c.sync()
                for k in range(3):
                    got = c.to_host(bufs[k], cap)
                    if not np.array_equal(got, want[k]):
                        bad += 1
                        d = np.flatnonzero(got != want[k])
                        print(f"MISMATCH seq {seq} kind {kind} op {op} buffer {k}: {d.size} bytes, first {d[:3]}", flush=True)
                        want[k] = got.copy()                 # (report once)
                continue
            if r < 0.24 and kind in (0, 2):                  # a new source image / a new mesh in the middle of the queue (both settle queued runs)
                u = rng.random()
                if u < 0.3:
                    imgs = [G.lcg_image(W, H, 7000 + seq * 16 + op)]
                    img = imgs[0]
                    c.set_image(img)
                elif u < 0.6:                                # one source per frame: frame f reads image f mod n_images
                    imgs = [G.lcg_image(W, H, 9000 + seq * 16 + op + q) for q in range(int(rng.integers(2, 4)))]
                    img = imgs[0]
                    c.sync()
                    if d_srcs:
                        c.free(d_srcs)
                    d_srcs = c.alloc(W * H * 4 * len(imgs))
                    for q, im in enumerate(imgs):
                        c.to_device(d_srcs, im, q * W * H * 4)
                    c.set_images_device(d_srcs, W, H, len(imgs), W * H * 4)
                else:
                    nx, ny = int(rng.integers(1, 10)), int(rng.integers(1, 8))
                    sp = WL.grid_points(W, H, nx, ny).reshape(-1, 2).astype(np.float64)
                    tris = WL.grid_triangles(nx, ny)
                    sp32 = sp.astype(np.float32).ravel()
                    ms = O.minmax_xy(sp32)
                    fmap_w, fmap_h = int(ms[2] - ms[0]), int(ms[3] - ms[1])
                    fmap = O.build_tri_map(sp32, tris, fmap_w, int(ms[1]), fmap_w * fmap_h) if fmap_w > 0 and fmap_h > 0 else None
                    c.piecewise_set_mesh(sp32, tris, int(ms[0]), int(ms[1]))
                continue
            if r < 0.3:
                c.set_option(str(rng.choice(["phase", "patch", "tri_group", "xcc_rotate", "self_spans", "tile", "min_row_groups"])), int(rng.choice([-1, 0, 1, 2])))
                continue
            k = int(rng.integers(0, 3))
            n = int(rng.integers(1, 4))
            frames = [new_points() for _ in range(n)]
            geoms = [geom_of(d) for d in frames]
            if any(g[2] <= 0 or g[3] <= 0 for g in geoms):
                continue
            offs, total = HG.pack_offsets(geoms)
            if total > cap:
                continue
            calls += 1
            if r < 0.38 and len(imgs) == 1:                  # an affine frame (inverse geometric: no pending record of its own) into the same pool
                a = rng.uniform(-0.5, 0.5); sc = rng.uniform(0.6, 1.4)
                m = np.array([np.cos(a) * sc, np.sin(a) * sc, -np.sin(a) * sc, np.cos(a) * sc, rng.uniform(-10, 30), rng.uniform(-10, 30), 0, 0], np.float64)
                lim = O.transform_limits(0, m[:6], W, H)
                g = tuple(int(v) for v in lim)
                if g[2] <= 0 or g[3] <= 0 or g[2] * g[3] * 4 > cap:
                    continue
                inv = np.asarray(HG.invert_affine(m[:6].astype(np.float32)), np.float64)
                c.warp_inverse_geometric_device(0, np.concatenate([inv, [0, 0]]), g, bufs[k])
                want[k][:g[2] * g[3] * 4] = O.warp_inverse_geometric(0, np.concatenate([inv, [0, 0]]), img, *g).ravel()
                continue
            if r < 0.65 or fmap is None or len(imgs) > 1:    # inverse batch (the forward warps take one source image)
                c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                c.warp_inverse_piecewise_frames_device(bufs[k])
                for f, g in enumerate(geoms):
                    w = O.warp_inverse_piecewise(sp32, frames[f], tris, imgs[f % len(imgs)], int(ms[0]), int(ms[1]), *g)
                    want[k][offs[f]:offs[f] + g[2] * g[3] * 4] = w.ravel()
            else:                                            # forward batch
                c.warp_forward_piecewise_batch_device(np.concatenate(frames), int(ms[2]), int(ms[3]), geoms, offs, bufs[k])
                for f, g in enumerate(geoms):
                    fwd = O.piecewise_matrices(sp32, frames[f], tris)
                    w = O.warp_forward_piecewise(fmap, fwd, img, int(ms[0]), int(ms[1]), int(ms[2]), int(ms[3]), *g)
                    want[k][offs[f]:offs[f] + g[2] * g[3] * 4] = w.ravel()
        redone += c.redone_frames()
        for b in bufs:
            c.free(b)
        if d_srcs:
            c.free(d_srcs)
    finally:
        c.close()
print(f"sequence fuzz done: {n_seq} sequences, {calls} batches, {redone} frames redone, {bad} mismatching buffers", flush=True)
sys.exit(1 if bad else 0)
