"""Experiment: k_pw_rows time vs row width at constant pixel count (per-row fixed cost vs per-window cost)."""
import sys, numpy as np
sys.path.insert(0, 'tests')
from hgtest import hip, workloads as WL, golden as G
HG = hip.load()
ctx = HG.Context(0)
H = 2160
for W in (960, 1920, 2304, 3840, 7680):
    F = max(1, int(32 * 3840 / W))
    img = G.lcg_image(W, H, 3)
    sp, tris = WL.grid_points(W, H, 10, 10), WL.grid_triangles(10, 10)
    frames = [WL.sin_dst(sp, 40.0, 8 + (f % 4)) for f in range(F)]
    geoms = [WL.piecewise_geom(d) for d in frames]
    ms = WL.src_min(sp)
    ctx.set_image(img)
    ctx.piecewise_set_mesh(sp, tris, ms[0], ms[1])
    offs, total = HG.pack_offsets(geoms)
    d = ctx.alloc(total)
    ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
    for _ in range(5): ctx.warp_inverse_piecewise_frames_device(d)
    ctx.sync(); ctx.set_timing(True)
    for _ in range(50): ctx.warp_inverse_piecewise_frames_device(d)
    ctx.sync()
    tot, n = ctx.kernel_ms_stats(); ctx.set_timing(False)
    px = sum(g[2] * g[3] for g in geoms); rows = sum(g[3] for g in geoms); nwin = sum(g[3] * ((g[2] + 255) // 256) for g in geoms)
    k = tot / n
    print(f"W={W} F={F} px={px/1e6:.1f}M rows={rows} windows={nwin} kernel={k:.4f} ms  ns/window={k*1e6/nwin:.4f}  Gpx/s={px/k/1e6:.1f}")
    ctx.free(d)
