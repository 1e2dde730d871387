"""Experiment: a lens-distortion style mesh (regular fine grid, displacements of a few pixels) on 4K, 32 frames."""
import sys, numpy as np
sys.path.insert(0, 'tests')
from hgtest import hip, workloads as WL, golden as G
HG = hip.load()
W, H, F = 3840, 2160, 32
img = G.lcg_image(W, H, 1)
for nx, ny, A in ((64, 36, 3.0), (32, 18, 3.0), (96, 54, 2.0)):
    sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
    frames = [WL.sin_dst(sp, A, 8 + (f % 4)) for f in range(F)]
    geoms = [WL.piecewise_geom(d) for d in frames]
    ms = WL.src_min(sp)
    offs, total = HG.pack_offsets(geoms)
    line = f"grid {nx}x{ny} ({tris.size//3} tri) A={A}:"
    for name, opts in (("auto", {}), ("patch", {"patch": 1}), ("rows1", {"patch": 0, "min_row_groups": 1 << 30}), ("groups4", {"patch": 0, "min_row_groups": 0})):
        ctx = HG.Context(0)
        for k, v in opts.items(): ctx.set_option(k, v)
        ctx.set_image(img); ctx.piecewise_set_mesh(sp, tris, ms[0], ms[1])
        d = ctx.alloc(total)
        ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
        for _ in range(20): ctx.warp_inverse_piecewise_frames_device(d)
        ctx.sync(); ctx.set_timing(True)
        for _ in range(60): ctx.warp_inverse_piecewise_frames_device(d)
        ctx.sync()
        tot, n = ctx.kernel_ms_stats()
        line += f"  {name}: k{ctx.last_piecewise_kernel()} {tot/n:.4f}"
        ctx.free(d); ctx.close()
    print(line)
