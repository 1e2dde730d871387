#!/usr/bin/env python3
"""Census of what the DEFAULT layout policy picks (round-5 verdict item 6): for every piecewise frame set of
  * BASELINE's configs C3 / C4 / C5 and the experiment grids, with a shared source and with one source per frame, F in {1, 8, 64},
  * the reference README's benchmark grid (2 / 400 / 23 040 triangles on 200^2 / 400^2 / 800^2 outputs),
  * every inverse piecewise warp of the 144 golden cases,
the variant code of the kernel instantiation that ran (hg_last_piecewise_variant) and whether a frame was redone.  Prints one line per
(variant, count, examples); instantiations and option keys that never show up here and lost wherever they were tried are what gets deleted.
    python tools/census.py [--quick]        (GPU box; through gpurun)"""
import collections, importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "homography.js_amd", rel))
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m


KIND = {1: "k_pw_rows", 3: "k_pw_rows_s80", 4: "k_pw_patch", 5: "k_pw_tile", 6: "k_pw_fused", 8: "k_pw_patch<GLOBALREC>"}


def describe(code):
    if code == 0: return "(none)"
    k, rest = divmod(code, 100000)
    cap, rest = divmod(rest, 10000)
    ph, rest = divmod(rest, 1000)
    cmp_, rest = divmod(rest, 100)
    tap = rest >= 50
    rest -= 50 if tap else 0
    hib, self_ = divmod(rest, 10)
    return f"{KIND.get(k, k)} cap={'512' if cap else '256'} per_phase={ph} entries={'8B' if cmp_ else '32B'} bounds={'hi' if hib else 'f64'} self={self_}{' map-tap' if tap else ''}"


def main():
    import numpy as np, torch
    hg, wl = load("hgwarp", "hgwarp.py"), load("hg_workloads", "workloads.py")
    quick = "--quick" in sys.argv
    dev = torch.device("cuda", 0)
    seen = collections.OrderedDict()

    def note(code, what, redone):
        e = seen.setdefault(code, {"n": 0, "ex": [], "redone": 0})
        e["n"] += 1; e["redone"] += redone
        if len(e["ex"]) < 6: e["ex"].append(what)

    # ---- configs x source layout x F
    for config in (["C3", "C4", "C5"] if quick else ["C3", "C4", "C5", "G16", "G24", "G40", "G64", "T12x60", "T20x60", "C3flat", "C5flat"]):
        cfg = wl.CONFIGS[config]; W, H = cfg["W"], cfg["H"]
        img = torch.from_numpy(wl.lcg_image(W, H, 1)).to(dev)
        for F in (1, 8, 64):
            if config.startswith("C5") and F == 64: continue           # (8 x 8K frames is BASELINE's batch per GPU; 64 would be 17 GB of output)
            if cfg["kind"] == "face":
                sp = wl.face_mesh(W, H, cfg["landmarks"]); tris = hg.triangulate(sp); seq = wl.face_frames(sp, W, cfg["total_frames"]); frames = [seq[f] for f in range(F)]
            else:
                sp, tris = wl.grid_points(W, H, cfg["nx"], cfg["ny"]), wl.grid_triangles(cfg["nx"], cfg["ny"])
                frames = [wl.sin_grid_dst(W, H, cfg["nx"], cfg["ny"], cfg["A"], 8 + f % 4) for f in range(F)]
            geoms = [wl.piecewise_geom(d) for d in frames]
            msx, msy = wl.src_min(sp)
            offs, total = hg.pack_offsets(geoms)
            out = torch.empty(total, dtype=torch.uint8, device=dev)
            for sources in ("shared", "distinct"):
                n_img = 1 if sources == "shared" else min(F, 8)
                with hg.Context(0) as ctx:
                    if sources == "shared": ctx.set_image_device(img.data_ptr(), W, H)
                    else:
                        flat = img.reshape(-1).repeat(n_img)
                        ctx.set_images_device(flat.data_ptr(), W, H, n_img, W * H * 4)
                    ctx.piecewise_set_mesh(sp, tris, msx, msy)
                    ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                    ctx.warp_inverse_piecewise_frames_device(out.data_ptr())
                    ctx.sync()
                    note(ctx.last_piecewise_variant(), f"{config} F={F} {sources}", ctx.redone_frames())
            del out
    # ---- the reference README's grid (test/benchmark.js:50-190): 400 x 400 source, 1 x 1 / 10 x 20 / 96 x 120 cells, outputs 200^2 / 400^2 / 800^2
    W = H = 400
    img = wl.lcg_image(W, H, 1)
    for nx, ny in ((1, 1), (10, 20), (96, 120)):
        sp, tris = wl.grid_points(W, H, nx, ny), wl.grid_triangles(nx, ny)
        for size in (200, 400, 800):
            dp = (wl.sin_dst(sp, 4.0, 8).reshape(-1, 2) * np.float32(size / 400.0)).astype(np.float32).ravel()
            geom = wl.piecewise_geom(dp)
            msx, msy = wl.src_min(sp)
            with hg.Context(0) as ctx:
                ctx.set_image(img)
                ctx.piecewise_set_mesh(sp, tris, msx, msy)
                ctx.piecewise_prepare(dp, geom)
                ctx.warp_inverse_piecewise()
                note(ctx.last_piecewise_variant(), f"README {tris.size // 3} triangles -> {size}^2", ctx.redone_frames())
    # ---- golden cases (single frames through hg_piecewise_prepare, as the class's warp() issues them)
    from hgtest import golden as G
    gold = G.load()
    for case in gold["cases"]:
        for k, w in enumerate(case["warps"]):
            if not G.is_pixel_warp(w) or w.get("stale") or w["path"] != "_inversePiecewiseAffineWarp": continue
            if quick and w["objW"] * w["objH"] > (1 << 22): continue
            imgk = G.case_images(case)[G.warp_image_key(case, k)]
            sp, dp = G.f32_from_bits(w["srcPoints"]), G.f32_from_bits(w["dstPoints"])
            with hg.Context(0) as ctx:
                ctx.set_image(imgk)
                ctx.piecewise_set_mesh(sp, G.warp_triangles(case, w), w["minSrcX"], w["minSrcY"])
                ctx.piecewise_prepare(dp, (w["xOff"], w["yOff"], w["objW"], w["objH"]))
                ctx.warp_inverse_piecewise()
                note(ctx.last_piecewise_variant(), f"golden {case['name']}#{k}", ctx.redone_frames())
    print(f"{'variant':>8}  {'sets':>5} {'redone':>6}  what / examples")
    for code, e in sorted(seen.items(), key=lambda kv: -kv[1]["n"]):
        print(f"{code:>8}  {e['n']:>5} {e['redone']:>6}  {describe(code)}   e.g. {'; '.join(e['ex'])}")
    print(json.dumps({"variants": {str(c): e["n"] for c, e in seen.items()}}))


if __name__ == "__main__":
    main()
