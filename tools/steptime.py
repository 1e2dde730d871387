"""Experiment: C3 step time / k_pw_rows time vs who owns the stream and the buffers (library vs torch)."""
import sys, time, os, importlib.util, numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from hgtest import hip, workloads as WL, golden as G
HG = hip.load()
cfg = WL.CONFIGS["C3"]; W, H, F = cfg["W"], cfg["H"], 32
img = G.lcg_image(W, H, 1)
sp, tris = WL.grid_points(W, H, 10, 10), WL.grid_triangles(10, 10)
frames = [WL.sin_dst(sp, 40.0, 8 + (f % 4)) for f in range(F)]
geoms = [WL.piecewise_geom(d) for d in frames]
ms = WL.src_min(sp)
offs, total = HG.pack_offsets(geoms)
dev = torch.device("cuda", 0)
for tstream, tout, timg in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)):
    st = torch.cuda.Stream(device=dev) if tstream else None
    ctx = HG.Context(0, stream=st.cuda_stream) if tstream else HG.Context(0)
    if timg:
        it = torch.from_numpy(img).to(dev); ctx.set_image_device(it.data_ptr(), W, H)
    else:
        ctx.set_image(img)
    ctx.piecewise_set_mesh(sp, tris, ms[0], ms[1])
    if tout:
        ot = torch.empty(total, dtype=torch.uint8, device=dev); d = ot.data_ptr()
    else:
        d = ctx.alloc(total)
    ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
    ctx.set_timing(True)
    for _ in range(10): ctx.warp_inverse_piecewise_frames_device(d)
    ctx.sync(); ctx.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(200): ctx.warp_inverse_piecewise_frames_device(d)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 200 * 1e3
    tot, n = ctx.kernel_ms_stats()
    print(f"torch stream={tstream} out={tout} img={timg}: {dt:.4f} ms/step, kernel {tot/n:.4f} ms, out ptr {d:#x}")
    ctx.close()
