"""Synthetic workloads of BASELINE.json / SURVEY.md §8(d): point grids, destination point sets, output windows.

Shared by bench.py and the tests (load by path, like hgwarp.py).  Pure numpy, no GPU, no oracle.
All points are float32 in pixel coordinates (the reference's Float32Array with pointsAreNormalized=false)."""
import numpy as np


def js_round(x):
    """Math.round on arrays: ties toward +Infinity."""
    x = np.asarray(x, np.float64)
    r = np.floor(x)
    return r + ((x - r) >= 0.5)


def lcg_image(w, h, seed):
    """SURVEY.md §8d synthetic RGBA: s = s*1664525 + 1013904223 mod 2^32; byte = s >> 24 (vectorised jump-ahead)."""
    n = w * h * 4
    a, c, mask = np.uint64(1664525), np.uint64(1013904223), np.uint64(0xFFFFFFFF)
    mul = np.empty(n, np.uint64)
    add = np.empty(n, np.uint64)
    mul[0], add[0] = a, c
    filled = 1
    while filled < n:
        m = min(filled, n - filled)
        am, cm = mul[filled - 1], add[filled - 1]
        mul[filled:filled + m] = (mul[:m] * am) & mask
        add[filled:filled + m] = (mul[:m] * cm + add[:m]) & mask
        filled += m
    s = (mul * np.uint64(seed) + add) & mask
    return (s >> np.uint64(24)).astype(np.uint8).reshape(h, w, 4)


def grid_points(W, H, nx, ny):
    """(nx+1) x (ny+1) source points (i*W/nx, j*H/ny), row-major (j outer), flat x,y float32."""
    xs = np.arange(nx + 1, dtype=np.float64) * (W / nx)
    ys = np.arange(ny + 1, dtype=np.float64) * (H / ny)
    gx, gy = np.meshgrid(xs, ys)
    return np.stack([gx, gy], -1).astype(np.float32).ravel()


def grid_triangles(nx, ny):
    """Row-major split (a,b,c),(b,d,c): a=(i,j) b=(i+1,j) c=(i,j+1) d=(i+1,j+1).  (Delaunator's choice of diagonal on a
    regular grid is implementation-defined and its source is absent: triangulation parity is unpinned)"""
    stride = nx + 1
    i, j = np.meshgrid(np.arange(nx), np.arange(ny))
    a = (j * stride + i).ravel()
    t = np.stack([a, a + 1, a + stride, a + 1, a + stride + 1, a + stride], 1)
    return t.astype(np.uint32).ravel()


def sin_dst(src_pts, A, n=8):
    """dst = (x, A + y + sin(n*x/pi)*A): the sinusoidal pattern of README.md:69-73 / test/benchmark.js:68."""
    p = np.asarray(src_pts, np.float32).reshape(-1, 2).astype(np.float64)
    d = np.stack([p[:, 0], A + p[:, 1] + np.sin((n * p[:, 0]) / np.pi) * A], 1)
    return d.astype(np.float32).ravel()


def sin_grid_dst(W, H, nx, ny, A, n=8):
    """Destination points of the sinusoidal grid computed like the reference's harness does (and tests/golden/gen_golden.mjs):
    from the UNROUNDED double grid coordinates x = i*(W/nx), y = j*(H/ny), then stored as float32.  Differs from
    sin_dst(grid_points(...)) whenever W/nx is not a float32 value (C5: 153.6), where the f32-rounded x moves the sine."""
    xs = np.arange(nx + 1, dtype=np.float64) * (W / nx)
    ys = np.arange(ny + 1, dtype=np.float64) * (H / ny)
    gx, gy = np.meshgrid(xs, ys)
    d = np.stack([gx, A + gy + np.sin((n * gx) / np.pi) * A], -1)
    return d.astype(np.float32).ravel()


def piecewise_geom(dst_pts):
    """_induceBestObjectiveWidthAndHeight, piecewise / pixel-coordinate branch (:706-710): round each of min/max, then subtract."""
    p = np.asarray(dst_pts, np.float32).reshape(-1, 2).astype(np.float64)
    mn, mx = js_round(p.min(0)), js_round(p.max(0))
    return (int(mn[0]), int(mn[1]), int(mx[0] - mn[0]), int(mx[1] - mn[1]))


def src_min(src_pts):
    """_minSrcX/_minSrcY: rounded source-point bbox minimum (:758)."""
    p = np.asarray(src_pts, np.float32).reshape(-1, 2).astype(np.float64)
    mn = js_round(p.min(0))
    return int(mn[0]), int(mn[1])


def projective_dst(W, H, t=0.0):
    """test/benchmark.js:282-283 corner pattern, animated by t (moves the right edge like the harness' movementArray)."""
    return np.array([W / 10, 0, W / 10, H, W, H * (2 / 8 - t), W, H * (6 / 8 + t)], np.float32)


def corners(W, H):
    return np.array([0, 0, 0, H, W, 0, W, H], np.float32)


def affine_dst(W, H, t=0.0):
    """test/benchmark.js:204-205 pattern."""
    return np.array([0, H / 2, W / 2, H * (8 / 10 + t), W / 2, 0], np.float32)


# ------------------------------------------------------------------ BASELINE.json configs
CONFIGS = {
    "C1": dict(kind="affine", W=400, H=400),
    "C2": dict(kind="projective", W=1920, H=1080),
    "C3": dict(kind="piecewise", W=3840, H=2160, nx=10, ny=10, A=40.0),
    # 68-landmark face mesh, 512 frames in total; triangles from the host Delaunay (hg_triangulate)
    "C4": dict(kind="face", W=3840, H=2160, landmarks=68, total_frames=512),
    "C5": dict(kind="piecewise", W=7680, H=4320, nx=50, ny=50, A=80.0),
    # experiments only (EXPERIMENTS.md): the same meshes (almost) without vertical displacement
    "C3flat": dict(kind="piecewise", W=3840, H=2160, nx=10, ny=10, A=1.0),
    # C5's mesh density without its steep shear
    "C5flat": dict(kind="piecewise", W=7680, H=4320, nx=50, ny=50, A=8.0),
    # 4K grids between C3's 200 and C5's 5 000 triangles (producer / self-span policy, EXPERIMENTS.md)
    "G16": dict(kind="piecewise", W=3840, H=2160, nx=16, ny=16, A=24.0),
    "G24": dict(kind="piecewise", W=3840, H=2160, nx=24, ny=24, A=16.0),
    "G40": dict(kind="piecewise", W=3840, H=2160, nx=40, ny=40, A=10.0),
    "G64": dict(kind="piecewise", W=3840, H=2160, nx=64, ny=36, A=6.0),
    # sparse rows, many triangles (20 cells across, 60 down: 2 400 triangles, ~42 spans per row): beyond k_pw_rows<SELF>'s scan limit
    "T20x60": dict(kind="piecewise", W=3840, H=2160, nx=20, ny=60, A=10.0),
    "T12x60": dict(kind="piecewise", W=3840, H=2160, nx=12, ny=60, A=10.0),
}


def piecewise_frames(cfg, n_frames):
    """Source mesh + n_frames destination point sets: sin((8 + f mod 4) * x / pi) as in test/benchmark.js:68."""
    sp = grid_points(cfg["W"], cfg["H"], cfg["nx"], cfg["ny"])
    tris = grid_triangles(cfg["nx"], cfg["ny"])
    frames = [sin_grid_dst(cfg["W"], cfg["H"], cfg["nx"], cfg["ny"], cfg["A"], 8 + (f % 4)) for f in range(n_frames)]
    geoms = [piecewise_geom(d) for d in frames]
    return sp, tris, frames, geoms


def face_mesh(W, H, n_landmarks=68, seed=68):
    """C4: deterministic 68-landmark layout (seeded jitter inside the central 60 % box)."""
    rng = np.random.default_rng(seed)
    cols = int(np.ceil(np.sqrt(n_landmarks)))
    pts = []
    for k in range(n_landmarks):
        i, j = k % cols, k // cols
        x = (0.2 + 0.6 * (i + 0.5 + rng.uniform(-0.35, 0.35)) / cols) * W
        y = (0.2 + 0.6 * (j + 0.5 + rng.uniform(-0.35, 0.35)) / cols) * H
        pts.append((x, y))
    return np.array(pts, np.float32).ravel()


def face_frames(sp, W, n_frames):
    """C4 frame f: dst = src + 0.02*W*(sin(phi+k), cos(phi+2k)), phi = 2*pi*f/F."""
    p = np.asarray(sp, np.float32).reshape(-1, 2).astype(np.float64)
    k = np.arange(p.shape[0])
    out = []
    for f in range(n_frames):
        phi = 2 * np.pi * f / n_frames
        out.append((p + 0.02 * W * np.stack([np.sin(phi + k), np.cos(phi + 2 * k)], 1)).astype(np.float32).ravel())
    return out
