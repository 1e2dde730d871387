#!/usr/bin/env python3
"""Forward (scatter-semantics) geometric warp of a 4K source, F frames per call, outputs resident on the device:
tile-binned gather (k_fwd_tiles) vs scatter + gather.  python tools/bench_forward.py [F] [reps] [W H]"""
import importlib.util, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "homography.js_amd", rel))
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m

hg, wl = load("hgwarp", "hgwarp.py"), load("hg_workloads", "workloads.py")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
ONLY = os.environ.get("FWD_ONLY", "")             # substring filter on the case name
ctx = hg.Context(0)
ctx.set_image(wl.lcg_image(W, H, 1))
cases = {"affine rot 0.1 scale 1.0": (0, 0.1, 1.0, 0.0), "affine rot 0.6 scale 0.7": (0, 0.6, 0.7, 0.0), "affine rot -0.3 scale 1.5": (0, -0.3, 1.5, 0.0),
         "projective rot 0.1 persp 5e-5": (1, 0.1, 1.0, 5e-5)}
for name, (kind, ang, s, g) in cases.items():
    if ONLY not in name: continue
    mats, geoms = [], []
    for f in range(F):
        a = ang + 0.01 * f
        if kind == 0:
            m = np.array([np.cos(a) * s, np.sin(a) * s, -np.sin(a) * s, np.cos(a) * s, 10.0 * f, 5.0, 0, 0], np.float64)
        else:
            m = np.array([np.cos(a) * s, -np.sin(a) * s, 10.0 * f, np.sin(a) * s, np.cos(a) * s, 5.0, g, g / 2], np.float64)
        mats.append(m)
        geoms.append(tuple(int(v) for v in hg.transform_limits(kind, m[:6] if kind == 0 else m, W, H)))
    offs, total = hg.pack_offsets(geoms)
    d = ctx.alloc(total)
    px = sum(gm[2] * gm[3] for gm in geoms)
    res = {"case": name, "frames": F, "source": f"{W}x{H}", "out_Mpx_per_frame": round(px / F / 1e6, 2)}
    ref = None
    for mode, label in ((0, "scatter_gather"), (1, "tiles")):
        ctx.set_option("fwd_tiles", mode)
        for _ in range(3):
            ctx.warp_forward_geometric_batch_device(kind, np.concatenate(mats), geoms, offs, d)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.warp_forward_geometric_batch_device(kind, np.concatenate(mats), geoms, offs, d)
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        res[label + "_us_per_frame"] = round(dt / F * 1e6, 1)
        res[label + "_kernel"] = ctx.last_forward_kernel()
        out = ctx.to_host(d, total)
        if ref is None: ref = out
        else: res["same_bytes"] = bool(np.array_equal(ref, out))
    ctx.free(d)
    print(json.dumps(res), flush=True)

# piecewise forward: the BASELINE C3 mesh (10 x 10 cells) and a dense one, shrunk to fit the source size
for name, (gx, gy, A) in {"piecewise 10x10 grid": (10, 10, 40.0), "piecewise 32x18 grid": (32, 18, 16.0), "piecewise 48x27 grid": (48, 27, 10.0),
                          "piecewise 96x54 grid": (96, 54, 6.0)}.items():
    if ONLY not in name: continue
    sp, tris = wl.grid_points(W, H, gx, gy), wl.grid_triangles(gx, gy)
    msx, msy = wl.src_min(sp)
    mm = hg.minmax_xy(sp)
    frames = [(wl.sin_grid_dst(W, H, gx, gy, A, 8 + k % 4).reshape(-1, 2) * np.float32(0.9) + np.float32(20)).astype(np.float32).ravel() for k in range(F)]
    geoms = [wl.piecewise_geom(d) for d in frames]
    offs, total = hg.pack_offsets(geoms)
    ctx.piecewise_set_mesh(sp, tris, msx, msy)
    d = ctx.alloc(total)
    px = sum(gm[2] * gm[3] for gm in geoms)
    res = {"case": name, "frames": F, "source": f"{W}x{H}", "out_Mpx_per_frame": round(px / F / 1e6, 2)}
    ref = None
    for mode, label in ((0, "scatter_gather"), (1, "tiles")):
        ctx.set_option("fwd_tiles", mode)
        for _ in range(3):
            ctx.warp_forward_piecewise_batch_device(np.concatenate(frames), int(mm[2]), int(mm[3]), geoms, offs, d)
        ctx.sync()
        r0 = ctx.redone_frames()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.warp_forward_piecewise_batch_device(np.concatenate(frames), int(mm[2]), int(mm[3]), geoms, offs, d)
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        res[label + "_us_per_frame"] = round(dt / F * 1e6, 1)
        res[label + "_kernel"] = ctx.last_forward_kernel()
        res[label + "_redone"] = ctx.redone_frames() - r0
        out = ctx.to_host(d, total)
        if ref is None: ref = out
        else: res["same_bytes"] = bool(np.array_equal(ref, out))
    ctx.free(d)
    print(json.dumps(res), flush=True)
