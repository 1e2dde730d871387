#!/usr/bin/env python3
"""Wide fuzz of the HIP path against the CPU oracle (run on the GPU box): tools/fuzz_gpu.py [trials] [seed].
FUZZ_TILE=1: wide frames, every trial also as a 3-frame batch (shared source and one source per frame) with k_pw_tile forced half of the time.
FUZZ_ROWS=1: the same frames through the self-span row kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hgtest import golden as G, hip, oracle as O, workloads as WL  # noqa: E402

HG = hip.load()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = HG.Context(0)
ROWS = bool(os.environ.get("FUZZ_ROWS"))
TILE = bool(os.environ.get("FUZZ_TILE")) or ROWS
bad = 0
seen = {}                                                  # kernel id -> batch runs that went through it
overflow = 0
for t in range(trials):
    W, H = int(rng.integers(8, 700)), int(rng.integers(8, 400))
    if TILE: W, H = int(rng.integers(300, 1500)), int(rng.integers(24, 260))
    img = G.lcg_image(W, H, 9000 + t)
    mode = t % 8
    # a random kernel-layout policy per trial (results must not depend on it): k_pw_rows phases, row groups, k_pw_patch variants,
    # windows per wave of the geometric kernel
    ctx.set_option("phase", int(rng.choice([-1, 1, 2, 4])))
    ctx.set_option("patch", int(rng.choice([-1, -1, 0, 1, 2])))
    ctx.set_option("min_row_groups", int(rng.choice([1536, 0, 1 << 30])))
    ctx.set_option("geo_windows", int(rng.choice([1, 2, 4, 8])))
    ctx.set_option("fwd_tiles", int(rng.choice([-1, 0, 1, 1])))
    ctx.set_option("hi_bounds", int(rng.choice([1, 1, 0])))
    ctx.set_option("self_spans", int(rng.choice([-1, 1, 0])))
    ctx.set_option("tile", int(rng.choice([-1, 1, 1, 0])))
    ctx.set_option("safe_spans", int(rng.choice([-1, 1, 0])))
    if TILE:
        ctx.set_option("self_spans", 1); ctx.set_option("patch", 0 if ROWS else 1); ctx.set_option("min_row_groups", 0)
    ctx.set_option("xcc_rotate", int(rng.choice([-1, 0, 1])))
    ctx.set_option("tri_group", int(rng.choice([-1, 0, 16, 64])))
    ctx.set_option("compact", int(rng.choice([-1, -1, 0, 1])))
    ctx.set_option("xcc", int(rng.choice([8, 8, 1, 2, 4, 16])))
    nx, ny = int(rng.integers(1, 24)), int(rng.integers(1, 16))
    if mode == 6:
        nx, ny = int(rng.integers(30, 140)), int(rng.integers(1, 6))                      # dense rows: 1 row per workgroup, > 63 spans per row
    sp = WL.grid_points(W, H, nx, ny).reshape(-1, 2).astype(np.float64)
    tris = WL.grid_triangles(nx, ny)
    if mode == 1:
        sp = sp * rng.uniform(0.3, 0.9) + rng.uniform(0, 0.3) * np.array([W, H])        # minSrc > 0
    elif mode == 2:
        sp = sp * 1.3 - np.array([W, H]) * 0.15                                            # minSrc < 0 (general kernel)
    jit = rng.uniform(0, 0.45)
    dp = (sp + rng.uniform(-jit, jit, sp.shape) * [W / nx, H / ny]) * rng.uniform(0.3, 3.0, 2) + rng.uniform(-80, 120, 2)
    if mode == 3 or (TILE and t % 2):
        dp[:, 1] += np.sin(dp[:, 0] * 0.37) * rng.uniform(1, 40)                            # steep shear
    if mode == 4:
        dp = np.round(dp * 2) / 2                                                           # .0 / .5 vertices (ties)
    if mode == 5:
        perm = rng.permutation(tris.size // 3)                                              # shuffled triangle order + a fold
        tris = tris.reshape(-1, 3)[perm].ravel()
        dp[rng.integers(0, dp.shape[0])] += rng.uniform(-60, 60, 2)
    sp32, dp32 = sp.astype(np.float32).ravel(), dp.astype(np.float32).ravel()
    ms, md = O.minmax_xy(sp32), O.minmax_xy(dp32)
    geom = (int(md[0]), int(md[1]), int(md[2] - md[0]), int(md[3] - md[1]))
    if geom[2] <= 0 or geom[3] <= 0 or geom[2] * geom[3] > 6_000_000:
        continue
    want, wmap, wfwd, winv = O.warp_inverse_piecewise(sp32, dp32, tris, img, int(ms[0]), int(ms[1]), *geom, taps=True)
    ctx.set_image(img)
    ctx.piecewise_set_mesh(sp32, tris, int(ms[0]), int(ms[1]))
    ctx.piecewise_prepare(dp32, geom)
    got = ctx.warp_inverse_piecewise()
    ok = np.array_equal(got, want) and np.array_equal(ctx.get_tri_map(), wmap)
    try:
        ok = ok and np.array_equal(ctx.get_tri_map(fused=True), wmap)
    except HG.HgError:
        overflow += 1          # more spans in a row than the fused kernel's LDS list: the warp itself went through the map path
    if mode == 7 or TILE:                                                                   # the same mesh as a 3-frame batch with different windows
        frames = [dp32, (dp * rng.uniform(0.6, 1.4, 2) + rng.uniform(-30, 30, 2)).astype(np.float32).ravel(),
                  (dp + rng.uniform(-3, 3, dp.shape)).astype(np.float32).ravel()]
        geoms = []
        for d in frames:
            m = O.minmax_xy(d)
            geoms.append((int(m[0]), int(m[1]), int(m[2] - m[0]), int(m[3] - m[1])))
        if all(g[2] > 0 and g[3] > 0 and g[2] * g[3] < 6_000_000 for g in geoms):
            offs, total = HG.pack_offsets(geoms)
            d_out = ctx.alloc(total)
            ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
            ctx.warp_inverse_piecewise_frames_device(d_out)
            ctx.warp_inverse_piecewise_frames_device(d_out)                                 # queued twice: self-cleaning counters
            ctx.sync()
            seen[ctx.last_piecewise_kernel()] = seen.get(ctx.last_piecewise_kernel(), 0) + 1
            if os.environ.get("FUZZ_DEBUG"): print("batch", t, mode, "kernel", ctx.last_piecewise_kernel(), "self", ctx.last_piecewise_self(), "geoms", geoms, "tris", tris.size // 3, "redone", ctx.redone_frames(), "flag", hex(ctx.last_piecewise_flag()), flush=True)
            for f, g in enumerate(geoms):
                gotf = ctx.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
                ok = ok and np.array_equal(gotf, O.warp_inverse_piecewise(sp32, frames[f], tris, img, int(ms[0]), int(ms[1]), *g))
            # ... and with one source per frame (hg_set_images_device)
            imgs = [img, G.lcg_image(W, H, 5000 + t), G.lcg_image(W, H, 7000 + t)]
            d_src = ctx.alloc(W * H * 4 * 3)
            for k3 in range(3):
                ctx.to_device(d_src, imgs[k3], k3 * W * H * 4)
            ctx.set_images_device(d_src, W, H, 3, W * H * 4)
            ctx.warp_inverse_piecewise_frames_device(d_out)
            ctx.sync()
            seen[ctx.last_piecewise_kernel()] = seen.get(ctx.last_piecewise_kernel(), 0) + 1
            for f, g in enumerate(geoms):
                gotf = ctx.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
                ok = ok and np.array_equal(gotf, O.warp_inverse_piecewise(sp32, frames[f], tris, imgs[f], int(ms[0]), int(ms[1]), *g))
            ctx.set_image(img)
            ctx.free(d_src)
            ctx.free(d_out)
    # geometric kernels on the same image
    d4 = (WL.corners(W, H).reshape(4, 2) * rng.uniform(0.4, 2.0, 2) + rng.uniform(-0.2, 0.2, (4, 2)) * [W, H] + rng.uniform(-50, 50, 2)).astype(np.float32).ravel()
    s4 = WL.corners(W, H)
    fw = O.projective_from_squares(s4, d4)
    lim = O.transform_limits(1, fw, W, H)
    if np.all(np.isfinite(lim)) and 0 < lim[2] * lim[3] < 6_000_000:
        lim = [int(v) for v in lim]
        inv = O.projective_from_squares(d4, s4)
        wantg = O.warp_inverse_geometric(1, inv, img, *lim)
        ok = ok and np.array_equal(ctx.warp_inverse_geometric(1, HG.solve_projective(d4, s4), lim), wantg)
        if t % 3 == 0:                                                                      # the same frame with the matrix solved on the device
            d_g = ctx.alloc(lim[2] * lim[3] * 4)
            ctx.geometric_set_frames_points(1, d4, s4, [lim])
            ctx.warp_inverse_geometric_frames_device(d_g)
            ctx.sync()
            ok = ok and np.array_equal(ctx.to_host(d_g, lim[2] * lim[3] * 4).reshape(lim[3], lim[2], 4), wantg)
            ok = ok and np.array_equal(ctx.get_geometric_matrices(1)[0].view(np.uint64), inv.view(np.uint64))
            ctx.free(d_g)
    fa = O.affine_from_triangles(s4[:6], d4[:6]).astype(np.float64)
    lim = O.transform_limits(0, fa, W, H)
    if np.all(np.isfinite(lim)) and 0 < lim[2] * lim[3] < 6_000_000:
        lim = [int(v) for v in lim]
        ia = O.affine_from_triangles(d4[:6], s4[:6]).astype(np.float64)
        ok = ok and np.array_equal(ctx.warp_inverse_geometric(0, ia, lim), O.warp_inverse_geometric(0, ia, img, *lim))
    # forward (scatter) paths: what warp() takes for same-size outputs; wider than the unit test (windows that cut the image,
    # magnifications with holes, strong shear, negative offsets)
    if t % 4 == 0:
        s3 = np.array([0, 0, 0, H, W, 0], np.float32)
        d3 = (s3.reshape(3, 2) * rng.uniform(0.3, 1.8, 2) + rng.uniform(-40, 40, 2) + rng.uniform(-25, 25, (3, 2))).astype(np.float32).ravel()
        fa = O.affine_from_triangles(s3, d3).astype(np.float64)
        lim = O.transform_limits(0, fa, W, H)
        if np.all(np.isfinite(lim)) and 0 < lim[2] * lim[3] < 3_000_000:
            lim = [int(v) for v in lim]
            ok = ok and np.array_equal(ctx.warp_forward_geometric(0, fa, lim), O.warp_forward_geometric(0, fa, img, *lim))
        fp = O.projective_from_squares(s4, d4)
        lim = O.transform_limits(1, fp, W, H)
        if np.all(np.isfinite(lim)) and 0 < lim[2] * lim[3] < 3_000_000:
            lim = [int(v) for v in lim]
            ok = ok and np.array_equal(ctx.warp_forward_geometric(1, fp, lim), O.warp_forward_geometric(1, fp, img, *lim))
        mw, mh = int(ms[2] - ms[0]), int(ms[3] - ms[1])
        if 0 < mw * mh < 3_000_000 and geom[2] * geom[3] < 3_000_000:
            fwdm = O.piecewise_matrices(sp32, dp32, tris)
            fmap = O.build_tri_map(sp32, tris, mw, int(ms[1]), mw * mh)
            wantf = O.warp_forward_piecewise(fmap, fwdm, img, int(ms[0]), int(ms[1]), int(ms[2]), int(ms[3]), *geom)
            ctx.piecewise_set_mesh(sp32, tris, int(ms[0]), int(ms[1]))
            ok = ok and np.array_equal(ctx.warp_forward_piecewise(dp32, int(ms[2]), int(ms[3]), geom), wantf)
    if not ok:
        bad += 1
        print("MISMATCH trial", t, "mode", mode, W, H, nx, ny, geom, flush=True)
print(f"fuzz done: {trials} trials, {bad} mismatches, {overflow} frames through the map-path fallback, batch runs by kernel id {dict(sorted(seen.items()))}, redone frames {ctx.redone_frames()}")
sys.exit(1 if bad else 0)
