#!/usr/bin/env python3
"""Upper bound for pipelining the span producer of step i+1 under the warp kernel of step i: TWO contexts on one device, each with
its own stream and buffers, enqueued alternately (A, B, A, B, ...) against ONE context doing the same number of steps.
python tools/two_ctx.py CONFIG[,CONFIG..] [--sources shared|distinct] [--frames F]"""
import importlib.util, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "homography.js_amd", rel))
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m

def main():
    import numpy as np, torch
    hg, wl = load("hgwarp", "hgwarp.py"), load("hg_workloads", "workloads.py")
    args = sys.argv[1:]
    configs = args[0].split(",")
    src = args[args.index("--sources") + 1] if "--sources" in args else "shared"
    Fopt = int(args[args.index("--frames") + 1]) if "--frames" in args else None
    dev = torch.device("cuda", 0)
    for config in configs:
        cfg = wl.CONFIGS[config]; W, H = cfg["W"], cfg["H"]
        F = Fopt or {"C5": 8}.get(config, 64)
        img = torch.from_numpy(wl.lcg_image(W, H, 1)).to(dev)
        if cfg["kind"] == "face":
            sp = wl.face_mesh(W, H, cfg["landmarks"]); tris = hg.triangulate(sp); seq = wl.face_frames(sp, W, cfg["total_frames"])
            frames = [seq[f] for f in range(F)]
        else:
            sp, tris = wl.grid_points(W, H, cfg["nx"], cfg["ny"]), wl.grid_triangles(cfg["nx"], cfg["ny"])
            frames = [wl.sin_grid_dst(W, H, cfg["nx"], cfg["ny"], cfg["A"], 8 + f % 4) for f in range(F)]
        geoms = [wl.piecewise_geom(d) for d in frames]
        msx, msy = wl.src_min(sp)
        offs, total = hg.pack_offsets(geoms)
        srcs = img.unsqueeze(0).repeat(F, 1, 1, 1) if src == "distinct" else None
        ctxs, outs = [], []
        for k in range(2):
            stream = torch.cuda.Stream(device=dev)
            c = hg.Context(0, stream=stream.cuda_stream)
            if srcs is not None: c.set_images_device(srcs.data_ptr(), W, H, F, W * H * 4)
            else: c.set_image_device(img.data_ptr(), W, H)
            c.piecewise_set_mesh(sp, tris, msx, msy)
            c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
            ctxs.append((c, stream)); outs.append(torch.zeros(total, dtype=torch.uint8, device=dev))     # (zeros: the packing leaves padding between frames that no kernel writes)
        res = {"config": config, "F": F, "sources": src}
        for mode in ("one", "two", "one", "two"):
            n = 100
            for rep in range(2):                              # warm, then timed
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    k = (i & 1) if mode == "two" else 0
                    ctxs[k][0].warp_inverse_piecewise_frames_device(outs[k].data_ptr())
                for c, _ in ctxs: c.sync()
                dt = (time.perf_counter() - t0) / n * 1e3
            res.setdefault(mode + "_ms_per_step", []).append(round(dt, 4))
        res["same_bytes"] = bool(torch.equal(outs[0], outs[1]))
        print(json.dumps(res), flush=True)
        for c, _ in ctxs: c.close()

if __name__ == "__main__":
    main()
