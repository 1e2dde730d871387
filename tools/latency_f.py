#!/usr/bin/env python3
"""Single-launch latency of small frame sets (F = 1, 2, 4, 8 4K frames of C3, device-resident in and out).
   python tools/latency_f.py            host view: ms per step with a sync after every step, and queued back to back
   rocprofv3 --kernel-trace -d DIR -o t -- python tools/latency_f.py --trace ; python tools/latency_f.py --parse DIR
                                        device view: k_tri_spans / k_tri_setup start -> warp kernel end of the synced steps, and each kernel alone"""
import glob, importlib.util, json, os, sqlite3, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "homography.js_amd", rel))
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m

FS = (1, 2, 4, 8)
REPS = 40

def run(trace):
    import numpy as np, torch
    hg, wl = load("hgwarp", "hgwarp.py"), load("hg_workloads", "workloads.py")
    cfg = wl.CONFIGS[os.environ.get("HG_LAT_CONFIG", "C3")]; W, H = cfg["W"], cfg["H"]
    dev = torch.device("cuda", 0)
    img = torch.from_numpy(wl.lcg_image(W, H, 1)).to(dev)
    sp, tris = wl.grid_points(W, H, cfg["nx"], cfg["ny"]), wl.grid_triangles(cfg["nx"], cfg["ny"])
    ms = wl.src_min(sp)
    for F in FS:
        frames = [wl.sin_grid_dst(W, H, cfg["nx"], cfg["ny"], cfg["A"], 8 + f % 4) for f in range(F)]
        geoms = [wl.piecewise_geom(d) for d in frames]
        offs, total = hg.pack_offsets(geoms)
        out = torch.empty(total, dtype=torch.uint8, device=dev)
        stream = torch.cuda.Stream(device=dev)
        ctx = hg.Context(0, stream=stream.cuda_stream)
        for kv in sys.argv[1:]:
            if "=" in kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        ctx.set_image_device(img.data_ptr(), W, H)
        ctx.piecewise_set_mesh(sp, tris, ms[0], ms[1])
        ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
        for _ in range(20): ctx.warp_inverse_piecewise_frames_device(out.data_ptr())
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(REPS):
            ctx.warp_inverse_piecewise_frames_device(out.data_ptr()); ctx.sync()
        synced = (time.perf_counter() - t0) / REPS * 1e3
        if not trace:
            t0 = time.perf_counter()
            for _ in range(REPS * 5): ctx.warp_inverse_piecewise_frames_device(out.data_ptr())
            ctx.sync()
            queued = (time.perf_counter() - t0) / (REPS * 5) * 1e3
            print(json.dumps({"F": F, "kernel": ctx.last_piecewise_kernel(), "host_ms_per_synced_step": round(synced, 4), "ms_per_queued_step": round(queued, 4),
                              "us_per_frame_queued": round(queued * 1e3 / F, 2)}), flush=True)
        ctx.close()

def parse(d):
    db = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))[0]
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = next((c for c in ("kernel_name", "name", "kernel") if c in cols), None)
    if name is None:
        print("kernels view columns:", cols); return
    rows = con.execute(f"select {name}, start, end from kernels where {name} like '%hg::%' order by start").fetchall()
    steps = []                                             # (tri_start, tri_end, warp_start, warp_end)
    i = 0
    while i + 1 < len(rows):
        if ("k_tri_spans" in rows[i][0] or "k_tri_setup" in rows[i][0]) and (any(k in rows[i + 1][0] for k in ("k_pw_rows", "k_pw_patch", "k_pw_tile"))):
            steps.append((rows[i][1], rows[i][2], rows[i + 1][1], rows[i + 1][2], rows[i + 1][0])); i += 2
        else: i += 1
    per = 20 + REPS                                        # warmup + synced steps per F
    for k, F in enumerate(FS):
        s = steps[k * per + 20: (k + 1) * per]
        if not s: continue
        med = lambda v: sorted(v)[len(v) // 2]
        print(json.dumps({"F": F, "kernel": s[0][4].split("<")[0].split("::")[-1], "steps": len(s),
                          "tri_spans_us": round(med([a[1] - a[0] for a in s]) / 1e3, 2), "gap_us": round(med([a[2] - a[1] for a in s]) / 1e3, 2),
                          "warp_us": round(med([a[3] - a[2] for a in s]) / 1e3, 2), "step_us": round(med([a[3] - a[0] for a in s]) / 1e3, 2),
                          "us_per_frame": round(med([a[3] - a[0] for a in s]) / 1e3 / F, 2)}))

if __name__ == "__main__":
    if "--parse" in sys.argv: parse(sys.argv[sys.argv.index("--parse") + 1])
    else: run("--trace" in sys.argv)
