"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/hgwarp.h declares, its host-side
solves are bit-exact against the golden vectors, and it refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from hgtest import golden as G
from hgtest import hip

ROOT = hip.ROOT
HG = hip.load()
GOLD = G.load()


def _declared():
    text = open(os.path.join(ROOT, "include", "hgwarp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 30
    L = C.CDLL(HG.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(HG.EXPORTS) == names          # the ctypes binding covers the whole header
    assert HG.lib().hg_version() == 100


def _same(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    bits = np.uint32 if a.dtype == np.float32 else np.uint64
    eq = a.view(bits) == b.view(bits)
    return bool(np.all(eq | (np.isnan(a) & np.isnan(b)) | ((a == 0) & (b == 0))))


def test_host_solves_bit_exact_against_reference_vectors():
    f = GOLD["func"]
    for v in f["affine"]:
        assert _same(HG.solve_affine(G.f32_from_bits(v["src"]), G.f32_from_bits(v["dst"])), G.f32_from_bits(v["out"]))
    for v in f["inv_affine"]:
        assert _same(HG.invert_affine(G.f32_from_bits(v["m"])), G.f32_from_bits(v["out"]))
    for v in f["projective"]:
        assert _same(HG.solve_projective(G.f32_from_bits(v["src"]), G.f32_from_bits(v["dst"])), G.f64_from_hex(v["out"]))
    for v in f["limits"]:
        if v["kind"] == "affine":
            got = HG.transform_limits(0, G.f32_from_bits(v["m"]).astype(np.float64), v["w"], v["h"])
        else:
            got = HG.transform_limits(1, G.f64_from_hex(v["m"]), v["w"], v["h"])
        assert _same(got, G.f64_from_hex(v["out"]))
    for v in f["minmax"]:
        assert _same(HG.minmax_xy(G.f32_from_bits(v["p"])), G.f64_from_hex(v["out"]))
    for v in f["round"]:
        x, want = G.f64_from_hex([v["x"]])[0], G.f64_from_hex([v["r"]])[0]
        got = HG.js_round(x)
        assert (np.isnan(got) and np.isnan(want)) or got == want


def test_register_lu_equals_oracle_lu_on_random_and_degenerate_squares():
    """hg_solve_projective is the fully unrolled, conditional-swap restatement that also runs per frame on the GPU
    (solve_projective_regs); the oracle keeps numeric.js' loop form with row-pointer swaps.  Same bits on 20 000 seeded
    squares incl. ones that force every pivot choice, repeated points (singular systems: Inf / NaN) and huge ranges."""
    from hgtest import oracle as O
    rng = np.random.default_rng(1234)
    n_nan = 0
    for trial in range(20000):
        kind = trial % 5
        if kind == 0:
            s = rng.uniform(-2000, 4000, 8); d = rng.uniform(-2000, 4000, 8)
        elif kind == 1:                                      # near-axis-aligned rectangles (many exact zeros / ties in the pivot search)
            w, h = rng.integers(1, 4000, 2)
            s = np.array([0, 0, 0, h, w, 0, w, h], float); d = s + rng.integers(-3, 4, 8)
        elif kind == 2:                                      # normalised coordinates
            s = rng.uniform(0, 1, 8); d = rng.uniform(0, 1, 8)
        elif kind == 3:                                      # repeated / collinear points
            s = rng.integers(0, 4, 8).astype(float); d = rng.integers(0, 4, 8).astype(float)
        else:                                                # wide dynamic range
            s = rng.uniform(-1, 1, 8) * 10.0 ** rng.integers(-6, 7, 8); d = rng.uniform(-1, 1, 8) * 10.0 ** rng.integers(-6, 7, 8)
        s32, d32 = s.astype(np.float32), d.astype(np.float32)
        got, want = HG.solve_projective(s32, d32), O.projective_from_squares(s32, d32)
        n_nan += int(np.isnan(want).any())
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)) or (np.isnan(got) == np.isnan(want)).all() and _same(got, want), (trial, s32, d32)
    assert n_nan > 100                                       # the degenerate branch was exercised


def test_multi_device_partition_covers_every_frame_once():
    """hg_multi_partition (the frame -> device map of hg_multi_warp_piecewise_batch): contiguous blocks, every frame owned
    exactly once, sizes differ by at most one; the same split as dist.shard_frames of the torch.distributed path."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("hg_dist", os.path.join(ROOT, "homography.js_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    for n in (0, 1, 7, 8, 64, 511, 512, 513):
        for G in (1, 2, 3, 8):
            blocks = [HG.multi_partition(n, G, k) for k in range(G)]
            flat = [f for first, cnt in blocks for f in range(first, first + cnt)]
            assert flat == list(range(n))
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
            assert [list(range(a, a + c)) for a, c in blocks] == [list(D.shard_frames(n, k, G)) for k in range(G)]
    with pytest.raises(HG.HgError):
        HG.multi_partition(8, 0, 0)
    with pytest.raises(HG.HgError):
        HG.multi_partition(8, 2, 2)


def test_multi_device_fanout_plan_falls_back_per_pair():
    """hg_multi_plan_fanout (pure function): the copy plan of the shared-source fan-out.  Full peer access: scatter of 1/G slices
    from device 0, then an all-gather in which every ordered pair of devices carries exactly ONE slice (xGMI is point-to-point: no
    link carries more than 1/G of the image) and the pulls of one step use different sources.  A pair without peer access re-routes
    only ITS copies: to the root if the reader reaches it, else to the host buffer; every device still ends with every slice."""
    for G in (1, 2, 3, 4, 8):
        full = np.ones((G, G), np.uint8)
        ops = HG.multi_plan_fanout(full)
        assert len(ops) == (G - 1) + (G - 1) * (G - 1)
        assert [o for o in ops if o[3] == 0] == [(q, 0, q, 0) for q in range(1, G)]            # scatter first, in order
        gather = [o for o in ops if o[3] == 1]
        assert ops[:G - 1] == [o for o in ops if o[3] == 0]                                      # ... before any all-gather copy
        assert all(src == sl for _, src, sl, _ in gather)                                        # every slice from its owner
        links = {}
        for dst, src, sl, ph in ops:
            links[(src, dst)] = links.get((src, dst), 0) + 1
        assert all(v == 1 for (s, d), v in links.items() if s != 0), links                       # peer links: one slice each
        assert all(links.get((0, q), 0) == (2 if G > 1 else 0) for q in range(1, G))             # root link: scatter slice q + slice 0
        for q in range(1, G):                                                                     # device q ends up with every slice once
            assert sorted(sl for dst, _, sl, _ in ops if dst == q) == list(range(G))
        for k in range(G - 2):                                                                    # step k of the all-gather: distinct sources
            step = [[o for o in gather if o[0] == q][k][1] for q in range(1, G)]
            assert len(set(step)) == len(step), (G, k, step)
    # one missing pair, both directions: 2 <-> 3 of 4
    acc = np.ones((4, 4), np.uint8)
    acc[2, 3] = acc[3, 2] = 0
    ops = HG.multi_plan_fanout(acc)
    assert (3, 0, 2, 1) in ops and (2, 0, 3, 1) in ops                                           # slice 2 for device 3 (and 3 for 2) from the root
    assert not any((src, dst) in ((2, 3), (3, 2)) for dst, src, _, _ in ops)
    assert sum(1 for dst, src, sl, ph in ops if src == -1) == 0
    # device 3 cannot reach the root either: its own slice and slice 0 come from the host, the others from their owners
    acc = np.ones((4, 4), np.uint8)
    acc[0, 3] = 0
    ops = HG.multi_plan_fanout(acc)
    assert (3, -1, 3, 0) in ops and (3, -1, 0, 1) in ops and (3, 1, 1, 1) in ops and (3, 2, 2, 1) in ops
    # no peer access at all: every device but the root reads the whole image from the host
    none = np.eye(4, dtype=np.uint8)
    ops = HG.multi_plan_fanout(none)
    assert all(src == -1 for _, src, _, _ in ops)
    for q in range(1, 4):
        assert sorted(sl for dst, _, sl, _ in ops if dst == q) == [0, 1, 2, 3]
    assert HG.lib().hg_multi_plan_fanout(0, none.ctypes.data, None, 0) < 0


def test_pack_offsets():
    offs, total = HG.pack_offsets([(0, 0, 10, 3), (5, -2, 0, 7), (0, 0, 64, 64)])
    assert offs == [0, 256, 256] and total == 256 + 64 * 64 * 4


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(HG.HgError) as e:
        HG.Context(0)
    assert "no CPU fallback" in str(e.value) or "device" in str(e.value)


def _circumcircle_violations(P, tris):
    P = P.astype(np.float64)
    bad = 0
    for a, b, c in tris.reshape(-1, 3):
        A, B, Cc = P[a], P[b], P[c]
        d = 2 * (A[0] * (B[1] - Cc[1]) + B[0] * (Cc[1] - A[1]) + Cc[0] * (A[1] - B[1]))
        assert abs(d) > 1e-12, "degenerate triangle"
        ux = ((A @ A) * (B[1] - Cc[1]) + (B @ B) * (Cc[1] - A[1]) + (Cc @ Cc) * (A[1] - B[1])) / d
        uy = ((A @ A) * (Cc[0] - B[0]) + (B @ B) * (A[0] - Cc[0]) + (Cc @ Cc) * (B[0] - A[0])) / d
        r2 = (A[0] - ux) ** 2 + (A[1] - uy) ** 2
        d2 = (P[:, 0] - ux) ** 2 + (P[:, 1] - uy) ** 2
        d2[[a, b, c]] = np.inf
        bad += int((d2 < r2 * (1 - 1e-9)).sum())
    return bad


def test_triangulate_is_delaunay_and_matches_scipy_on_generic_points():
    """hg_triangulate stands where the reference calls Delaunator (:1216).  No reference test pins the triangle order
    (SURVEY.md §8c), so the bar is: valid Delaunay, and the same triangle SET as an independent implementation."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(5)
    for n in (3, 4, 7, 68, 300):
        P = (rng.random((n, 2)) * [1000, 700]).astype(np.float32)
        t = HG.triangulate(P.ravel())
        assert t.dtype == np.uint32 and t.size % 3 == 0 and t.size > 0
        assert _circumcircle_violations(P, t) == 0
        mine = {tuple(sorted(x)) for x in t.reshape(-1, 3).tolist()}
        ref = {tuple(sorted(x)) for x in Delaunay(P.astype(np.float64)).simplices.tolist()}
        assert mine == ref
    # scale: the same triangle set as Qhull on 20 000 points, in well under a second (adjacency-based insertion)
    P = (rng.random((20000, 2)) * [4000, 3000]).astype(np.float32)
    t = HG.triangulate(P.ravel())
    assert {tuple(sorted(x)) for x in t.reshape(-1, 3).tolist()} == {tuple(sorted(x)) for x in Delaunay(P.astype(np.float64)).simplices.tolist()}
    # co-circular inputs: a lattice is covered exactly, a circle + centre (+ duplicates) gives the fan
    g = np.stack(np.meshgrid(np.arange(60) * 7.0, np.arange(40) * 5.0), -1).reshape(-1, 2).astype(np.float32)
    tri = g[HG.triangulate(g.ravel()).reshape(-1, 3)].astype(np.float64)
    area = 0.5 * np.abs((tri[:, 1, 0] - tri[:, 0, 0]) * (tri[:, 2, 1] - tri[:, 0, 1]) - (tri[:, 2, 0] - tri[:, 0, 0]) * (tri[:, 1, 1] - tri[:, 0, 1]))
    assert tri.shape[0] == 2 * 59 * 39 and area.min() > 0 and abs(area.sum() - 59 * 7 * 39 * 5) < 1e-6
    th = np.linspace(0, 2 * np.pi, 500, endpoint=False)
    circ = np.stack([np.cos(th), np.sin(th)], 1) * 1000 + 1500
    circ = np.vstack([circ, [[1500, 1500]], circ[:50]]).astype(np.float32)
    assert HG.triangulate(circ.ravel()).size // 3 == 500
    assert HG.triangulate(np.array([0, 0, 1, 1], np.float32)).size == 0                 # fewer than 3 points
    assert HG.triangulate(np.array([0, 0, 1, 1, 2, 2], np.float32)).size == 0           # collinear
    with pytest.raises(HG.HgError):
        HG.triangulate(np.array([0, 0, 1, np.nan, 2, 2], np.float32))


def test_c4_face_mesh_workload():
    """C4: 68 deterministic landmarks, own Delaunay (~120 triangles), rotating offsets per frame."""
    from hgtest import workloads as WL
    W, H = 3840, 2160
    sp = WL.face_mesh(W, H)
    tris = HG.triangulate(sp)
    assert sp.size == 136 and 100 <= tris.size // 3 <= 130
    assert _circumcircle_violations(sp.reshape(-1, 2), tris) == 0
    frames = WL.face_frames(sp, W, 8)
    assert len(frames) == 8 and all(f.dtype == np.float32 and f.size == 136 for f in frames)
    assert np.abs(frames[0].reshape(-1, 2) - sp.reshape(-1, 2)).max() <= 0.02 * W + 1e-3


def test_projective_plain_range_classification():
    """Which projective frames may use the shared-reciprocal division (host-side proof, no GPU): ordinary homographies yes;
    a horizon (denominator changing sign) inside the window, non-finite or denormal-scale entries, absurd windows no."""
    g = (0, 0, 1920, 1080)
    ident = [1, 0, 0, 0, 1, 0, 0, 0]
    assert HG.projective_plain_range(ident, g)
    assert HG.projective_plain_range([0.9, 0.1, -30, -0.05, 1.1, 12, 1e-4, -2e-4], g)             # den in [0.78, 1.19]
    assert not HG.projective_plain_range([1, 0, 0, 0, 1, 0, -1e-3, 0], g)                          # den = 1 - x/1000: zero at x = 1000
    assert HG.projective_plain_range([1, 0, 0, 0, 1, 0, -1e-3, 0], (0, 0, 700, 1080))              # same matrix, window ends before the horizon
    assert not HG.projective_plain_range([1, 0, 0, 0, 1, 0, -1e-3, 0], (0, 0, 800, 1080))          # (the ragged last 256-pixel window is computed too)
    assert not HG.projective_plain_range([1, 0, 0, 0, 1, 0, float("nan"), 0], g)
    assert not HG.projective_plain_range([1, 0, 0, 0, 1e-200, 0, 0, 0], g)
    assert not HG.projective_plain_range([1e150, 0, 0, 0, 1, 0, 0, 0], g)
    assert not HG.projective_plain_range(ident, (1 << 28, 0, 100, 100))
    assert HG.projective_plain_range(ident, (0, 0, 0, 0))                                          # empty window: nothing to divide


def test_one_fma_form_is_admitted_only_where_it_has_the_reference_bits():
    """hg_affine_one_fma_form (the predicate k_tri_setup applies per triangle): wherever it says yes, fma(m0, x, (m2*y) + m4) -- evaluated
    here in exact rational arithmetic, rounded once -- equals the reference's ((m0*x) + (m2*y)) + m4 (two roundings, plain doubles) for
    pixels of the window, incl. its corners; it says yes for ordinary meshes (C3's matrices, face-mesh-like matrices) and no for non-finite
    entries, shears 2^-40 of the scale, and windows at 2^24."""
    from fractions import Fraction
    rng = np.random.default_rng(11)

    def reference(m0, m2, m4, x, y):
        return ((float(m0) * x) + (float(m2) * y)) + float(m4)          # :1383: products exact in doubles, two rounded sums

    def one_fma(m0, m2, m4, x, y):
        c = (float(m2) * y) + float(m4)                                  # the record's A
        return float(Fraction(float(m0)) * x + Fraction(c))              # fma = the exact sum rounded once (Fraction -> float rounds to nearest even)

    yes = 0
    for trial in range(3000):
        kind = trial % 6
        m = rng.standard_normal(6).astype(np.float32)
        if kind == 1: m *= np.float32(2.0) ** rng.integers(-30, 12, 6).astype(np.float32)   # wildly different magnitudes
        if kind == 2: m[[0, 3]] = 1; m[[1, 2]] *= np.float32(2.0 ** -int(rng.integers(0, 45)))
        if kind == 3: m[rng.integers(0, 6)] = 0
        if kind == 4: m[4:] *= 4000
        if kind == 5: m = np.array([1, rng.normal() * 0.2, 0, 1, 0, rng.normal() * 200], np.float32)      # C3-like
        g = (int(rng.integers(-50, 50)), int(rng.integers(-50, 50)), int(rng.integers(1, 8192)), int(rng.integers(1, 4500)))
        if not HG.affine_one_fma_form(m, g):
            continue
        yes += 1
        xs = [g[0], g[0] + g[2] - 1] + [int(v) for v in rng.integers(g[0], g[0] + g[2], 6)]
        ys = [g[1], g[1] + g[3] - 1] + [int(v) for v in rng.integers(g[1], g[1] + g[3], 6)]
        for x in xs:
            for y in ys:
                assert one_fma(m[0], m[2], m[4], x, y) == reference(m[0], m[2], m[4], x, y), (m, g, x, y)
                assert one_fma(m[1], m[3], m[5], x, y) == reference(m[1], m[3], m[5], x, y), (m, g, x, y)
    assert yes > 1200                                                    # the predicate is not vacuous
    g4k = (0, 1, 3840, 2239)
    assert HG.affine_one_fma_form(np.array([1, 0.0756, -0.0, 1, -0.0, -40.0], np.float32), g4k)
    assert HG.affine_one_fma_form(np.array([0.97, 0.11, -0.08, 1.04, 35.5, -12.25], np.float32), g4k)
    assert not HG.affine_one_fma_form(np.array([1, 0, 2.0 ** -40, 1, 0, 0], np.float32), g4k)        # m0 x + m2 y needs 12 + 40 + 24 bits
    assert not HG.affine_one_fma_form(np.array([1, 0, np.nan, 1, 0, 0], np.float32), g4k)
    assert not HG.affine_one_fma_form(np.array([np.inf, 0, 0, 1, 0, 0], np.float32), g4k)
    assert HG.affine_one_fma_form(np.zeros(6, np.float32), g4k)                                        # all sums are 0


def test_forward_tile_admission_bounds():
    """Host-side admission of the tile-binned forward kernel (hg_forward_tiles_admissible): ordinary matrices pass with a trusted
    inverse; anything that could put a source pixel far outside the window, a projective denominator near zero, huge or
    non-finite entries, narrow windows -> the scatter path; a singular matrix is admissible but without inverse."""
    W, H = 640, 480
    rot = lambda a, s=1.0, tx=0.0, ty=0.0: np.array([np.cos(a) * s, np.sin(a) * s, -np.sin(a) * s, np.cos(a) * s, tx, ty], np.float64)
    for a, s in ((0.0, 1.0), (0.3, 0.5), (-2.0, 2.5), (np.pi / 4, 1.0)):
        m = rot(a, s, 12.0, -7.0)
        lim = [int(v) for v in HG.transform_limits(0, m, W, H)]
        assert HG.forward_tiles_admissible(0, m, W, H, lim) == 2, (a, s)
        # the window moved 100 columns to the right: source pixels land 100 columns outside on the left
        assert HG.forward_tiles_admissible(0, m, W, H, (lim[0] + 100, lim[1], lim[2], lim[3])) == 0
        # ... moved by 20: inside the 30-column aliasing margin
        assert HG.forward_tiles_admissible(0, m, W, H, (lim[0] + 20, lim[1], lim[2] - 40, lim[3])) == 2
    m = rot(0.2)
    lim = [int(v) for v in HG.transform_limits(0, m, W, H)]
    assert HG.forward_tiles_admissible(0, m, W, H, (lim[0], lim[1], 40, lim[3])) == 0                     # narrower than two aliasing margins
    bad = m.copy(); bad[4] = np.nan
    assert HG.forward_tiles_admissible(0, bad, W, H, lim) == 0
    bad = m.copy(); bad[0] = 1e7
    assert HG.forward_tiles_admissible(0, bad, W, H, lim) == 0
    sing = np.array([1.0, 0.5, 2.0, 1.0, 0.0, 0.0])                                                      # rank 1: every pixel lands on one line
    lim = [int(v) for v in HG.transform_limits(0, sing, W, H)]
    if lim[2] >= 64 and lim[3] > 0:
        assert HG.forward_tiles_admissible(0, sing, W, H, lim) == 1
    p = np.array([1.0, 0.02, 3.0, -0.01, 0.9, 5.0, 1e-4, -5e-5])
    lim = [int(v) for v in HG.transform_limits(1, p, W, H)]
    assert HG.forward_tiles_admissible(1, p, W, H, lim) == 2
    p2 = p.copy(); p2[6] = -1.0 / 300.0                                                                   # denominator crosses zero inside the image
    assert HG.forward_tiles_admissible(1, p2, W, H, lim) == 0
    p3 = p.copy(); p3[6] = 0.5
    assert HG.forward_tiles_admissible(1, p3, W, H, lim) == 0
    assert HG.forward_tiles_admissible(0, m, 70000, H, lim) == 0                                           # beyond the 65 535 source limit
