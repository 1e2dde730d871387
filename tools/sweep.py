#!/usr/bin/env python3
"""Kernel-time sweep over hg_set_option knobs:  python tools/sweep.py CONFIG[,CONFIG..] key=v1,v2,.. [key=...] [--sources shared,distinct] [--frames F]
Prints the mean hipEvent duration of the dominant kernel and of the whole step for every combination."""
import importlib.util, itertools, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "homography.js_amd", rel))
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m

def main():
    import numpy as np, torch
    hg, wl = load("hgwarp", "hgwarp.py"), load("hg_workloads", "workloads.py")
    args = sys.argv[1:]
    configs = args[0].split(",")
    knobs, sources, Fopt = {}, ["shared", "distinct"], None
    i = 1
    while i < len(args):
        if args[i] == "--sources": sources = args[i + 1].split(","); i += 2
        elif args[i] == "--frames": Fopt = int(args[i + 1]); i += 2
        else: k, v = args[i].split("="); knobs[k] = [int(x) for x in v.split(",")]; i += 1
    dev = torch.device("cuda", 0)
    for config in configs:
        cfg = dict(wl.CONFIGS[config]); W, H = cfg["W"], cfg["H"]
        if "HG_A" in os.environ and "A" in cfg: cfg["A"] = float(os.environ["HG_A"])      # the sine's amplitude (shear experiments)
        F = Fopt or {"C5": 8, "C5flat": 8}.get(config, 64)
        img = torch.from_numpy(wl.lcg_image(W, H, 1)).to(dev)
        srcs = None
        proj = cfg["kind"] == "projective"
        if proj:
            s4 = wl.corners(W, H)
            d4s = [wl.projective_dst(W, H, 0.0125 * (f % 10)) for f in range(F)]
            geoms = [tuple(int(v) for v in hg.transform_limits(1, hg.solve_projective(s4, d4), W, H)) for d4 in d4s]
        elif cfg["kind"] == "face":
            sp = wl.face_mesh(W, H, cfg["landmarks"]); tris = hg.triangulate(sp); seq = wl.face_frames(sp, W, cfg["total_frames"])
            frames = [seq[f] for f in range(F)]
        else:
            sp, tris = wl.grid_points(W, H, cfg["nx"], cfg["ny"]), wl.grid_triangles(cfg["nx"], cfg["ny"])
            frames = [wl.sin_grid_dst(W, H, cfg["nx"], cfg["ny"], cfg["A"], 8 + f % 4) for f in range(F)]
        if not proj:
            geoms = [wl.piecewise_geom(d) for d in frames]
            msx, msy = wl.src_min(sp)
        offs, total = hg.pack_offsets(geoms)
        shift = int(os.environ.get("OUT_SHIFT", "0"))     # output base moved by this many bytes inside a larger allocation (placement experiments)
        out_alloc = torch.empty(total + shift, dtype=torch.uint8, device=dev)
        out = out_alloc[shift:]
        ref = None
        for src in sources:
            for combo in itertools.product(*knobs.values()):
                stream = torch.cuda.Stream(device=dev)
                ctx = hg.Context(0, stream=stream.cuda_stream)
                for k, v in zip(knobs, combo): ctx.set_option(k, v)
                if src == "distinct":
                    pad = int(os.environ.get("SRC_PAD", "0"))     # bytes between consecutive sources (HBM channel-mapping experiments)
                    if srcs is None:
                        flat = torch.empty(F * (W * H * 4 + pad), dtype=torch.uint8, device=dev)
                        for f in range(F): flat[f * (W * H * 4 + pad): f * (W * H * 4 + pad) + W * H * 4] = img.reshape(-1)
                        srcs = flat
                    ctx.set_images_device(srcs.data_ptr(), W, H, F, W * H * 4 + pad)
                else:
                    ctx.set_image_device(img.data_ptr(), W, H)
                if proj:
                    ctx.geometric_set_frames_points(1, np.concatenate(d4s), np.tile(s4, F), geoms, offs)
                    run = ctx.warp_inverse_geometric_frames_device
                else:
                    ctx.piecewise_set_mesh(sp, tris, msx, msy)
                    ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                    run = ctx.warp_inverse_piecewise_frames_device
                out.zero_(); torch.cuda.synchronize()
                for _ in range(60): run(out.data_ptr())
                ctx.sync()
                if ref is None: ref = out.clone()
                same = bool(torch.equal(ref, out))
                # two timed passes, the second reported: the first combination of a process reads 3-8 % slow otherwise (whatever the
                # combination: clocks / page placement settling after the reference clone above), which biased "baseline first" sweeps
                for rep in range(2):
                    ctx.set_timing(False); ctx.set_timing(True)
                    t0 = time.perf_counter()
                    for _ in range(100): run(out.data_ptr())
                    ctx.sync()
                    dt = (time.perf_counter() - t0) / 100 * 1e3
                    tot, n = ctx.kernel_ms_stats()
                print(json.dumps({"config": config, "F": F, "sources": src, **dict(zip(knobs, combo)), "kernel": ctx.last_piecewise_kernel(),
                                  "kernel_ms": round(tot / n, 4), "step_ms": round(dt, 4), "same_bytes": same, "redone": ctx.redone_frames()}), flush=True)
                ctx.close()

if __name__ == "__main__":
    main()
