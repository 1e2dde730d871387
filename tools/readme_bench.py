#!/usr/bin/env python3
"""The reference README's own benchmark grid (README.md:270-328: 400x400 RGBA source; affine, projective, piecewise with 2 /
360 / ~23 000 triangles; outputs of ~200^2, 400^2, 800^2) as steady-state `setDestinyPoints(dst); warp()` frames on the GPU:
ms per frame including the per-frame host work of the C ABI (point upload, estimates, launch, sync), output left in HBM."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hgtest import golden as G, hip, oracle as O, workloads as WL  # noqa: E402

HG = hip.load()
ctx = HG.Context(0)
W = H = 400
img = G.lcg_image(W, H, 1)
ctx.set_image(img)
published = {  # README.md:270-328, "rest of frames" ms on an i5-7500 / Chrome 92
    ("affine", 200): 0.7, ("affine", 400): 2.7, ("affine", 800): 10.8,
    ("projective", 200): 1.9, ("projective", 400): 7.2, ("projective", 800): 27.5,
    ("piecewise 2 tri", 200): 1.1, ("piecewise 2 tri", 400): 4.4, ("piecewise 2 tri", 800): 16.5,
    ("piecewise 360 tri", 200): 2.1, ("piecewise 360 tri", 400): 4.6, ("piecewise 360 tri", 800): 22.4,
    ("piecewise 22898 tri", 200): 24.3, ("piecewise 22898 tri", 400): 11.5, ("piecewise 22898 tri", 800): 62.0,
}

def timeit(f, n=200):
    for _ in range(40): f()                              # (past the one-time buffer growth and layout walk of a new mesh / window shape)
    best = None
    for _ in range(3):                                   # best of three passes: a pass that caught a host hiccup (ms-scale, seen on shared boxes) does not count
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(n): f()
        ctx.sync()
        dt = (time.perf_counter() - t0) / n * 1e3
        best = dt if best is None else min(best, dt)
    return best

rows = []
for size in (200, 400, 800):
    s = size / 400.0
    d_out = ctx.alloc(size * size * 4 * 2 + 4096)
    # affine / projective: destination = source corners scaled (plus a little skew so it is not the identity)
    s3 = np.array([0, 0, 0, H, W, 0], np.float32)
    d3 = np.array([0, 0, 4 * s, H * s, W * s, 6 * s], np.float32)
    inv = HG.solve_affine(d3, s3).astype(np.float64)
    geom = (0, 0, size, size)
    rows.append(("affine", size, timeit(lambda: (ctx.warp_inverse_geometric_device(0, inv, geom, d_out), ctx.sync()))))
    s4 = WL.corners(W, H)
    d4 = np.array([0, 0, 10 * s, H * s, W * s, 12 * s, W * s * 0.95, H * s * 0.97], np.float32)
    invp = HG.solve_projective(d4, s4)
    rows.append(("projective", size, timeit(lambda: (ctx.warp_inverse_geometric_device(1, invp, geom, d_out), ctx.sync()))))
    for nx in (1, 0, 107):                                   # 2 triangles; 360 (18 x 10 cells); 107 x 107 cells = 22 898
        gx, gy = (18, 10) if nx == 0 else (nx, nx)
        sp, tris = WL.grid_points(W, H, gx, gy), WL.grid_triangles(gx, gy)
        dst = [(WL.sin_dst(sp, 0.02 * H / gy, 8 + f).reshape(-1, 2) * np.float32(s)).astype(np.float32).ravel() for f in range(4)]
        geoms = [WL.piecewise_geom(d) for d in dst]
        ms = WL.src_min(sp)
        ctx.piecewise_set_mesh(sp, tris, ms[0], ms[1])
        k = [0]
        def frame():
            f = k[0] % 4; k[0] += 1
            ctx.piecewise_set_frames(dst[f], [geoms[f]], [0])   # == setDestinyPoints(dst_f)
            ctx.warp_inverse_piecewise_frames_device(d_out)      # == warp()
            ctx.sync()
        rows.append((f"piecewise {tris.size // 3} tri", size, timeit(frame, 100)))
    ctx.free(d_out)
print(f"{'experiment':24s} {'output':>9s} {'MI355X ms/frame':>16s} {'README ms (i5-7500, Chrome)':>28s} {'ratio':>8s}")
for name, size, ms in rows:
    pub = published.get((name, size))
    print(f"{name:24s} {size:4d}x{size:<4d} {ms:16.4f} {pub if pub else '-':>28} {('%.0fx' % (pub / ms)) if pub else '-':>8s}")
