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
ctx = hg.Context(0)
ctx.set_image(wl.lcg_image(W, H, 1))
cases = {"affine rot 0.1 scale 1.0": (0, 0.1, 1.0, 0.0), "affine rot 0.6 scale 0.7": (0, 0.6, 0.7, 0.0), "affine rot -0.3 scale 1.5": (0, -0.3, 1.5, 0.0),
         "projective rot 0.1 persp 5e-5": (1, 0.1, 1.0, 5e-5)}
for name, (kind, ang, s, g) in cases.items():
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
