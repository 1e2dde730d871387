#!/usr/bin/env python3
"""Timing ablations of k_pw_rows (experiments build only: `make -C homography.js_amd experiments`):
    python tools/ablate.py [C3|C4] [shared|distinct]      -> kernel ms for HG_ABLATE in {0, 2 no gathers, 4 no stores, 6 neither, 8 no search}
Each variant runs in its own process (the switch is read once).  Ablated variants write WRONG pixels: timing only."""
import importlib.util, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "homography.js_amd", rel))
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m

def one(config, sources, F):
    import numpy as np, torch
    hg, wl = load("hgwarp", "hgwarp.py"), load("hg_workloads", "workloads.py")
    hg.LIB_PATH = os.path.join(ROOT, "homography.js_amd", "lib", "libhgwarp_exp.so")
    cfg = wl.CONFIGS[config]; W, H = cfg["W"], cfg["H"]
    dev = torch.device("cuda", 0)
    img = torch.from_numpy(wl.lcg_image(W, H, 1)).to(dev)
    stream = torch.cuda.Stream(device=dev)
    ctx = hg.Context(0, stream=stream.cuda_stream)
    if cfg["kind"] == "face":
        sp = wl.face_mesh(W, H, cfg["landmarks"]); tris = hg.triangulate(sp); seq = wl.face_frames(sp, W, cfg["total_frames"])
        frames = [seq[f] for f in range(F)]
    else:
        sp, tris = wl.grid_points(W, H, cfg["nx"], cfg["ny"]), wl.grid_triangles(cfg["nx"], cfg["ny"])
        frames = [wl.sin_grid_dst(W, H, cfg["nx"], cfg["ny"], cfg["A"], 8 + f % 4) for f in range(F)]
    geoms = [wl.piecewise_geom(d) for d in frames]
    msx, msy = wl.src_min(sp)
    if sources == "distinct":
        srcs = img.unsqueeze(0).repeat(F, 1, 1, 1)
        ctx.set_images_device(srcs.data_ptr(), W, H, F, W * H * 4)
    else:
        ctx.set_image_device(img.data_ptr(), W, H)
    ctx.piecewise_set_mesh(sp, tris, msx, msy)
    offs, total = hg.pack_offsets(geoms)
    ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
    out = torch.empty(total, dtype=torch.uint8, device=dev)
    for _ in range(100): ctx.warp_inverse_piecewise_frames_device(out.data_ptr())
    ctx.sync(); ctx.set_timing(True)
    for _ in range(100): ctx.warp_inverse_piecewise_frames_device(out.data_ptr())
    ctx.sync()
    tot, n = ctx.kernel_ms_stats()
    print(json.dumps({"abl": int(os.environ.get("HG_ABLATE", "0")), "kernel_ms": round(tot / n, 4), "kernel": ctx.last_piecewise_kernel()}))

if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "--one":
        one(sys.argv[1], sys.argv[2], int(sys.argv[4]))
    else:
        config = sys.argv[1] if len(sys.argv) > 1 else "C3"; sources = sys.argv[2] if len(sys.argv) > 2 else "distinct"
        F = {"C5": 8}.get(config, 64)
        for abl in ([int(a) for a in sys.argv[3].split(',')] if len(sys.argv) > 3 else (0, 2, 4, 6, 8, 14)):
            p = subprocess.run([sys.executable, __file__, config, sources, "--one", str(F)], env=dict(os.environ, HG_ABLATE=str(abl)), capture_output=True, text=True, timeout=300)
            print(config, sources, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:])
