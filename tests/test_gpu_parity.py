"""GPU parity: the HIP path (through the C ABI) against (a) the golden vectors generated from the reference and
(b) the CPU oracle on the same inputs.  Bar: bit-exact RGBA, bit-exact triangle-index map, bit-exact f32 matrices.
Run with `pytest -m gpu` on an MI355X."""
import os
import sys

import numpy as np
import pytest

from hgtest import golden as G
from hgtest import hip
from hgtest import oracle as O
from hgtest import workloads as WL

pytestmark = pytest.mark.gpu

HG = hip.load()
GOLD = G.load()


# The piecewise fast path picks a kernel layout from the frame set (rows per workgroup, k_pw_patch for dense sheared meshes);
# results must not depend on it, so every test that takes `ctx` runs under each layout policy.
# ("hi_bounds": 0 keeps the fp64 form of the source-bounds tests, "xcc" changes the block id -> row band mapping: both are folded
#  into the existing layouts so that every kernel runs under either form without multiplying the suite.)
LAYOUTS = {"auto": {}, "groups4": {"min_row_groups": 0, "patch": 0, "hi_bounds": 0, "xcc": 4, "self_spans": 0, "tri_group": 1, "compact": 1, "safe_spans": 1}, "rows1": {"min_row_groups": 1 << 30, "patch": 0, "xcc": 1, "xcc_rotate": 0, "compact": 0},
           "patch": {"min_row_groups": 0, "patch": 1, "phase": 2, "tri_group": 64, "xcc_rotate": 0, "sub_bands": 2}, "patch_global": {"min_row_groups": 0, "patch": 2, "hi_bounds": 0, "xcc": 2},
           "phase1": {"phase": 1, "patch": 0, "geo_windows": 1, "fwd_tiles": 1, "hi_bounds": 0, "self_spans": 1, "safe_spans": 0, "xcc_rotate": 0, "sub_bands": 3},
           "phase4": {"phase": 4, "patch": 0, "min_row_groups": 0, "geo_windows": 2, "fwd_tiles": 0, "xcc": 16, "self_spans": 0, "xcc_rotate": 1, "tri_group": 0},
           "tile": {"min_row_groups": 0, "patch": 1, "self_spans": 1, "tile": 1, "xcc_rotate": 0, "sub_bands": 5}}


@pytest.fixture(scope="module", params=list(LAYOUTS))
def ctx(request):
    c = HG.Context(0)
    for k, v in LAYOUTS[request.param].items():
        c.set_option(k, v)
    yield c
    c.close()


def test_xcc_count_comes_from_the_device():
    """The row-band mapping is sized from hipDeviceAttributeNumberOfXccs, not from a constant: 8 on an unpartitioned MI355X
    (the only mode the test boxes run), and the option overrides it."""
    c = HG.Context(0)
    try:
        assert c.xcc_count() in (1, 2, 4, 8, 16), c.xcc_count()
        import torch
        props = torch.cuda.get_device_properties(0)
        if "gfx950" in props.gcnArchName and props.multi_processor_count == 256:
            assert c.xcc_count() == 8                              # unpartitioned MI355X: 8 XCDs x 32 CUs
        c.set_option("xcc", 2)
        assert c.xcc_count() == 2
        with pytest.raises(Exception):
            c.set_option("xcc", 3)
    finally:
        c.close()


def _inverse_warps():
    return [pytest.param(c["name"], k, id=f"{c['name']}#{k}") for c in GOLD["cases"] for k, w in enumerate(c["warps"]) if G.is_pixel_warp(w)]


def _nan_eq(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)) | ((a == 0) & (b == 0))))


def hip_run_warp(ctx, case, k, taps=False):
    w = case["warps"][k]
    img = G.case_images(case)[G.warp_image_key(case, k)]
    ctx.set_image(img)
    geom = (w["xOff"], w["yOff"], w["objW"], w["objH"])
    sp, dp = G.f32_from_bits(w["srcPoints"]), G.f32_from_bits(w["dstPoints"])
    if w.get("stale"):
        # the reference's loop read stale caches: the reference-state entry points get the matrices as the
        # reference held them and, for the forward loop, the definition of the map its shared field held
        fwd = G.blob(w["fwd"], np.float32).reshape(-1, 6)
        if w["path"] == "_inversePiecewiseAffineWarp":
            out = ctx.warp_inverse_piecewise_state(fwd, dp, G.warp_triangles(case, w), w["minSrcX"], w["minSrcY"], geom)
            return (out, None, None, (None, None), out) if taps else out
        pts, mtris, mw, mh, myoff = G.stale_map_def(w)
        return ctx.warp_forward_piecewise_state(fwd, pts, mtris, mw, mh, myoff, w["minSrcX"], w["minSrcY"], w["maxSrcX"], w["maxSrcY"], geom)
    if w["path"] == "_piecewiseAffineWarp":         # forward scatter (what warp() picks when the output is not larger)
        ctx.piecewise_set_mesh(sp, G.warp_triangles(case, w), w["minSrcX"], w["minSrcY"])
        return ctx.warp_forward_piecewise(dp, w["maxSrcX"], w["maxSrcY"], geom)
    if w["path"] == "_geometricWarp":
        kind = 0 if w["transform"] == "affine" else 1
        m = G.f32_from_bits(w["matrix"]["f32"]).astype(np.float64) if kind == 0 else G.f64_from_hex(w["matrix"]["f64"])
        return ctx.warp_forward_geometric(kind, m, geom)
    if w["transform"] == "piecewiseaffine":
        tris = G.warp_triangles(case, w)
        ctx.piecewise_set_mesh(sp, tris, w["minSrcX"], w["minSrcY"])
        ctx.piecewise_prepare(dp, geom)
        out = ctx.warp_inverse_piecewise()
        if not taps:
            return out
        return out, ctx.get_tri_map(), ctx.get_tri_map(fused=True), ctx.get_matrices(tris.size // 3), ctx.warp_inverse_piecewise_via_map()
    kind = 0 if w["transform"] == "affine" else 1
    # :994 the inverse matrix is re-solved from the swapped point sets (host-side solve of the same library)
    m = HG.solve_affine(dp, sp).astype(np.float64) if kind == 0 else HG.solve_projective(dp, sp)
    return ctx.warp_inverse_geometric(kind, m, geom)


@pytest.mark.parametrize("name,k", _inverse_warps())
def test_hip_matches_reference_golden(ctx, name, k):
    case = next(c for c in GOLD["cases"] if c["name"] == name)
    w = case["warps"][k]
    pw = w["path"] == "_inversePiecewiseAffineWarp"
    res = hip_run_warp(ctx, case, k, taps=pw)
    out = res[0] if pw else res
    assert out.shape == (w["out"]["h"], w["out"]["w"], 4)
    if "blob" in w["out"]:
        want = G.blob(w["out"]["blob"], np.uint8).reshape(out.shape)
        assert np.array_equal(out, want), f"{np.count_nonzero(np.any(out != want, axis=2))} pixels differ"
    assert G.sha256(out) == w["out"]["sha"]
    if pw and not w.get("stale"):
        _, map_k, map_f, (fwd, inv), out_map = res
        assert np.array_equal(out_map, out)                       # materialised-map path == fused path
        assert G.sha256(map_k) == w["map"]["sha"]                 # rasteriser kernel == reference Int16Array, bit-exact
        assert G.sha256(map_f) == w["map"]["sha"]                 # fused per-pixel lookup == reference map, bit-exact
        if "fwd" in w:
            assert _nan_eq(fwd.ravel(), G.blob(w["fwd"], np.float32))
            assert _nan_eq(inv.ravel(), G.blob(w["inv"], np.float32))
        if not np.isnan(fwd).any():
            assert G.sha256(fwd) == w["fwdSha"]
        if not np.isnan(inv).any():
            assert G.sha256(inv) == w["invSha"]


@pytest.mark.parametrize("name", [c["name"] for c in GOLD["cases"] if c["name"].startswith("fwd_")])
def test_full_size_forward_goldens_take_the_tile_kernels(name):
    """The reference's own forward warps at 1080p / 4K (tests/golden 5c): under the default policy they run the tile-binned
    kernels, forced off they run scatter + gather -- both give the reference's SHA-256."""
    case = next(c for c in GOLD["cases"] if c["name"] == name)
    c = HG.Context(0)
    try:
        for opt, kernel in ((-1, 2), (0, 1)):
            c.set_option("fwd_tiles", opt)
            r0 = c.redone_frames()
            out = hip_run_warp(c, case, 0)
            assert c.last_forward_kernel() == kernel and c.redone_frames() == r0, (name, opt, c.last_forward_kernel())
            assert G.sha256(out) == case["warps"][0]["out"]["sha"], (name, opt)
    finally:
        c.close()


def test_hip_matches_oracle_on_fresh_seeds(ctx):
    """Same seeded inputs through oracle and HIP at sizes the oracle finishes in well under a second."""
    rng = np.random.default_rng(1234)
    for trial in range(12):
        W, H = int(rng.integers(40, 300)), int(rng.integers(30, 200))
        img = G.lcg_image(W, H, 5000 + trial)
        nx, ny = int(rng.integers(1, 12)), int(rng.integers(1, 9))
        sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
        scale = rng.uniform(0.5, 2.2, 2)
        off = rng.uniform(-30, 60, 2)
        dp = (sp.reshape(-1, 2) + rng.uniform(-0.3, 0.3, (sp.size // 2, 2)) * [W / nx, H / ny]) * scale + off
        dp = dp.astype(np.float32).ravel()
        mm_s, mm_d = O.minmax_xy(sp), O.minmax_xy(dp)
        geom = (int(mm_d[0]), int(mm_d[1]), int(mm_d[2] - mm_d[0]), int(mm_d[3] - mm_d[1]))
        want, wmap, wfwd, winv = O.warp_inverse_piecewise(sp, dp, tris, img, int(mm_s[0]), int(mm_s[1]), *geom, taps=True)
        ctx.set_image(img)
        ctx.piecewise_set_mesh(sp, tris, int(mm_s[0]), int(mm_s[1]))
        ctx.piecewise_prepare(dp, geom)
        got = ctx.warp_inverse_piecewise()
        assert np.array_equal(got, want), (trial, np.count_nonzero(np.any(got != want, axis=2)))
        assert np.array_equal(ctx.get_tri_map(), wmap)
        assert np.array_equal(ctx.get_tri_map(fused=True), wmap)
        fwd, inv = ctx.get_matrices(tris.size // 3)
        assert _nan_eq(fwd, wfwd) and _nan_eq(inv, winv)
        # projective + affine on the same image
        d4 = np.array([[rng.uniform(0, .3) * W, rng.uniform(0, .3) * H], [rng.uniform(0, .3) * W, rng.uniform(.7, 1.5) * H],
                       [rng.uniform(.7, 1.5) * W, rng.uniform(0, .3) * H], [rng.uniform(.7, 1.5) * W, rng.uniform(.7, 1.5) * H]], np.float32).ravel()
        s4 = np.array([0, 0, 0, H, W, 0, W, H], np.float32)
        fwdm = O.projective_from_squares(s4, d4)
        lim = [int(v) for v in O.transform_limits(1, fwdm, W, H)]
        invm = O.projective_from_squares(d4, s4)
        assert np.array_equal(ctx.warp_inverse_geometric(1, HG.solve_projective(d4, s4), lim), O.warp_inverse_geometric(1, invm, img, *lim))
        fa = O.affine_from_triangles(s4[:6], d4[:6]).astype(np.float64)
        lim = [int(v) for v in O.transform_limits(0, fa, W, H)]
        ia = O.affine_from_triangles(d4[:6], s4[:6]).astype(np.float64)
        assert np.array_equal(ctx.warp_inverse_geometric(0, ia, lim), O.warp_inverse_geometric(0, ia, img, *lim))


def test_forward_scatter_matches_oracle(ctx):
    """_geometricWarp / _piecewiseAffineWarp (last writer in raster order wins) against the sequential oracle."""
    rng = np.random.default_rng(77)
    for trial in range(10):
        W, H = int(rng.integers(30, 200)), int(rng.integers(20, 150))
        img = G.lcg_image(W, H, 700 + trial)
        ctx.set_image(img)
        s3 = np.array([0, 0, 0, H, W, 0], np.float32)
        d3 = (s3.reshape(3, 2) * rng.uniform(0.4, 1.1, 2) + rng.uniform(-20, 30, 2) + rng.uniform(-10, 10, (3, 2))).astype(np.float32).ravel()
        fa = O.affine_from_triangles(s3, d3).astype(np.float64)
        lim = [int(v) for v in O.transform_limits(0, fa, W, H)]
        assert np.array_equal(ctx.warp_forward_geometric(0, fa, lim), O.warp_forward_geometric(0, fa, img, *lim)), ("affine", trial)
        s4 = np.array([0, 0, 0, H, W, 0, W, H], np.float32)
        d4 = (s4.reshape(4, 2) * rng.uniform(0.5, 1.0, 2) + rng.uniform(-5, 5, (4, 2))).astype(np.float32).ravel()
        fp = O.projective_from_squares(s4, d4)
        lim = [int(v) for v in O.transform_limits(1, fp, W, H)]
        assert np.array_equal(ctx.warp_forward_geometric(1, fp, lim), O.warp_forward_geometric(1, fp, img, *lim)), ("projective", trial)
        nx, ny = int(rng.integers(1, 7)), int(rng.integers(1, 6))
        sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
        if trial % 3 == 0:
            sp = (sp.reshape(-1, 2) * 0.8 + [W * 0.1 - 4, H * 0.1 + 3]).astype(np.float32).ravel()      # source bbox off the image corner
        dp = ((sp.reshape(-1, 2) + rng.uniform(-0.25, 0.25, (sp.size // 2, 2)) * [W / nx, H / ny]) * rng.uniform(0.5, 1.0, 2) + rng.uniform(0, 12, 2)).astype(np.float32).ravel()
        ms, md = O.minmax_xy(sp), O.minmax_xy(dp)
        geom = (int(md[0]), int(md[1]), int(md[2] - md[0]), int(md[3] - md[1]))
        mw, mh = int(ms[2] - ms[0]), int(ms[3] - ms[1])
        fwd = O.piecewise_matrices(sp, dp, tris)
        fmap = O.build_tri_map(sp, tris, mw, int(ms[1]), mw * mh)
        want = O.warp_forward_piecewise(fmap, fwd, img, int(ms[0]), int(ms[1]), int(ms[2]), int(ms[3]), *geom)
        ctx.piecewise_set_mesh(sp, tris, int(ms[0]), int(ms[1]))
        assert np.array_equal(ctx.warp_forward_piecewise(dp, int(ms[2]), int(ms[3]), geom), want), ("piecewise", trial)


def test_forward_tiles_match_scatter_and_oracle():
    """k_fwd_tiles (output tiles gather their source pixels, winners in LDS) == scatter + gather == the oracle's sequential
    loops, over rotations, up/down scaling, shear, perspective, windows that cut the image, negative offsets and windows
    shifted so that destination x lands just outside them (the flat-index aliasing into the neighbouring row)."""
    rng = np.random.default_rng(2024)
    c = HG.Context(0)
    try:
        took_tiles = 0
        for trial in range(60):
            W, H = int(rng.integers(100, 900)), int(rng.integers(80, 600))
            img = G.lcg_image(W, H, 4000 + trial)
            c.set_image(img)
            kind = trial % 2
            ang = rng.uniform(-3.2, 3.2) if trial % 3 else rng.choice([0.0, np.pi / 4, np.pi / 2, np.pi])
            sx, sy = rng.uniform(0.25, 3.0, 2) if trial % 5 else (1.0, 1.0)
            sh = rng.uniform(-0.6, 0.6) if trial % 4 == 0 else 0.0
            A = np.array([[np.cos(ang) * sx, -np.sin(ang) * sy + sh], [np.sin(ang) * sx, np.cos(ang) * sy]])
            t = rng.uniform(-300, 300, 2)
            if kind == 0:
                m = np.array([A[0, 0], A[1, 0], A[0, 1], A[1, 1], t[0], t[1]], np.float32).astype(np.float64)      # :1382 layout
            else:
                g = rng.uniform(-4e-4, 4e-4, 2) if trial % 7 else np.zeros(2)
                m = np.array([A[0, 0], A[0, 1], t[0], A[1, 0], A[1, 1], t[1], g[0], g[1]], np.float64)
            lim = O.transform_limits(kind, m, W, H)
            if not np.all(np.isfinite(lim)) or not (0 < lim[2] * lim[3] < 4_000_000):
                continue
            lim = [int(v) for v in lim]
            variants = [tuple(lim)]
            if lim[2] > 140 and lim[3] > 40:
                variants.append((lim[0] + 9, lim[1] - 3, lim[2] - 20, lim[3] + 5))         # x just outside on both sides: aliasing into neighbouring rows
                variants.append((lim[0] + lim[2] // 4, lim[1] + lim[3] // 3, lim[2] // 2, lim[3] // 2))   # window cuts the image: falls back to scatter
            for geom in variants:
                want = O.warp_forward_geometric(kind, m, img, *geom)
                c.set_option("fwd_tiles", 0)
                a = c.warp_forward_geometric(kind, m, geom)
                assert c.last_forward_kernel() == 1
                c.set_option("fwd_tiles", 1)
                b = c.warp_forward_geometric(kind, m, geom)
                took_tiles += c.last_forward_kernel() == 2
                assert np.array_equal(a, want), ("scatter", trial, geom)
                assert np.array_equal(b, want), ("tiles", trial, geom, c.last_forward_kernel())
        assert took_tiles >= 40, took_tiles
        # full size: 4K source, the policy picks the tile kernel by itself; batch of 3 frames in one launch == scatter path
        W, H = 3840, 2160
        img = G.lcg_image(W, H, 1)
        c.set_image(img)
        mats, geoms = [], []
        for f, (ang, s) in enumerate([(0.05, 1.0), (0.6, 0.7), (-0.3, 1.4)]):
            m6 = np.array([np.cos(ang) * s, np.sin(ang) * s, -np.sin(ang) * s, np.cos(ang) * s, 40.0 * f, -25.0], np.float32).astype(np.float64)
            mats.append(np.concatenate([m6, [0, 0]]))
            geoms.append(tuple(int(v) for v in O.transform_limits(0, m6, W, H)))
        offs, total = HG.pack_offsets(geoms)
        d_a, d_b = c.alloc(total), c.alloc(total)
        try:
            c.set_option("fwd_tiles", 0)
            c.warp_forward_geometric_batch_device(0, np.concatenate(mats), geoms, offs, d_a)
            c.sync()
            c.set_option("fwd_tiles", -1)
            c.warp_forward_geometric_batch_device(0, np.concatenate(mats), geoms, offs, d_b)
            c.sync()
            assert c.last_forward_kernel() == 2
            for f, g in enumerate(geoms):
                n = g[2] * g[3] * 4
                assert np.array_equal(c.to_host(d_a, n, offs[f]), c.to_host(d_b, n, offs[f])), f
            g = geoms[0]
            assert np.array_equal(c.to_host(d_b, g[2] * g[3] * 4, offs[0]).reshape(g[3], g[2], 4), O.warp_forward_geometric(0, mats[0][:6], img, *g))
        finally:
            c.free(d_a)
            c.free(d_b)
    finally:
        c.close()


def _fwd_pw_oracle(sp, dp, tris, img, geom):
    ms = O.minmax_xy(sp)
    mw, mh = int(ms[2] - ms[0]), int(ms[3] - ms[1])
    fwd = O.piecewise_matrices(sp, dp, tris)
    fmap = O.build_tri_map(sp, tris, mw, int(ms[1]), mw * mh)
    return O.warp_forward_piecewise(fmap, fwd, img, int(ms[0]), int(ms[1]), int(ms[2]), int(ms[3]), *geom)


def test_forward_piecewise_tiles_match_scatter_and_oracle():
    """The tile-binned forward piecewise warp (k_fwd_pw_bins + k_fwd_pw_tiles) == scatter + gather == the oracle's sequential
    loop: jittered / folded / shuffled meshes, source bbox off the image corner, degenerate source triangles (Inf / NaN matrices:
    the frame is flagged on the device and redone by hg_sync), tile lists that overflow (capacity grows), 4K frames and batches."""
    rng = np.random.default_rng(31337)
    c = HG.Context(0)
    try:
        tiles_taken = 0
        for trial in range(36):
            W, H = int(rng.integers(120, 700)), int(rng.integers(90, 500))
            img = G.lcg_image(W, H, 8000 + trial)
            c.set_image(img)
            nx, ny = int(rng.integers(1, 14)), int(rng.integers(1, 10))
            sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
            if trial % 3 == 0:
                sp = (sp.reshape(-1, 2) * 0.8 + [W * 0.1 - 4, H * 0.1 + 3]).astype(np.float32).ravel()      # min source x / y > 0
            jit = rng.uniform(0, 0.45)
            dp = ((sp.reshape(-1, 2) + rng.uniform(-jit, jit, (sp.size // 2, 2)) * [W / nx, H / ny]) * rng.uniform(0.5, 1.6, 2) + rng.uniform(-30, 40, 2))
            if trial % 4 == 1:
                dp[:, 1] += np.sin(dp[:, 0] * 0.21) * rng.uniform(1, 25)                                     # shear
            if trial % 5 == 2:
                perm = rng.permutation(tris.size // 3)
                tris = tris.reshape(-1, 3)[perm].ravel()
                dp[rng.integers(0, dp.shape[0])] += rng.uniform(-40, 40, 2)                                   # a fold
            if trial % 9 == 4:                                                                                # a degenerate source triangle: Inf / NaN matrices
                sp = sp.copy(); t0 = tris.reshape(-1, 3)[0]
                sp[2 * t0[1]:2 * t0[1] + 2] = sp[2 * t0[0]:2 * t0[0] + 2]
            dp = dp.astype(np.float32).ravel()
            md = O.minmax_xy(dp)
            geom = (int(md[0]), int(md[1]), int(md[2] - md[0]), int(md[3] - md[1]))
            if geom[2] <= 0 or geom[3] <= 0:
                continue
            ms = O.minmax_xy(sp)
            want = _fwd_pw_oracle(sp, dp, tris, img, geom)
            c.piecewise_set_mesh(sp, tris, int(ms[0]), int(ms[1]))
            c.set_option("fwd_tiles", 0)
            a = c.warp_forward_piecewise(dp, int(ms[2]), int(ms[3]), geom)
            assert c.last_forward_kernel() == 1
            c.set_option("fwd_tiles", 1)
            b = c.warp_forward_piecewise(dp, int(ms[2]), int(ms[3]), geom)
            tiles_taken += c.last_forward_kernel() == 2
            assert np.array_equal(a, want), ("scatter", trial)
            assert np.array_equal(b, want), ("tiles", trial, nx, ny, geom)
        assert tiles_taken >= 30, tiles_taken
        # tile lists that overflow their capacity: 12 800 triangles on 512 x 512 (about 200 entries per tile, first capacity 64)
        W = H = 512
        img = G.lcg_image(W, H, 77)
        c.set_image(img)
        sp, tris = WL.grid_points(W, H, 80, 80), WL.grid_triangles(80, 80)
        dp = (sp.reshape(-1, 2) * 0.9 + rng.uniform(-1.5, 1.5, (sp.size // 2, 2)) + 7).astype(np.float32).ravel()
        md, ms = O.minmax_xy(dp), O.minmax_xy(sp)
        geom = (int(md[0]), int(md[1]), int(md[2] - md[0]), int(md[3] - md[1]))
        want = _fwd_pw_oracle(sp, dp, tris, img, geom)
        c.piecewise_set_mesh(sp, tris, int(ms[0]), int(ms[1]))
        c.set_option("fwd_tiles", 1)
        redone0 = c.redone_frames()
        for rep in range(4):                                     # capacity 64 -> 128 -> 256: flagged + redone until the lists fit
            assert np.array_equal(c.warp_forward_piecewise(dp, int(ms[2]), int(ms[3]), geom), want), rep
        assert c.redone_frames() > redone0
        # 4K, the BASELINE mesh (200 triangles) shrunk to fit the source size (what warp() sends down the forward path): the policy
        # picks the tile kernels by itself; a batch of 3 frames == the scatter path, frame 0 == the oracle
        W, H = 3840, 2160
        img = G.lcg_image(W, H, 1)
        c.set_image(img)
        for (gx, gy) in ((10, 10), (96, 54)):
            sp, tris = WL.grid_points(W, H, gx, gy), WL.grid_triangles(gx, gy)
            ms = O.minmax_xy(sp)
            frames = [(WL.sin_grid_dst(W, H, gx, gy, 40.0 if gx == 10 else 6.0, 8 + k).reshape(-1, 2) * np.float32(0.9) + np.float32(20)).astype(np.float32).ravel() for k in range(3)]
            geoms = [WL.piecewise_geom(d) for d in frames]
            offs, total = HG.pack_offsets(geoms)
            c.piecewise_set_mesh(sp, tris, int(ms[0]), int(ms[1]))
            d_a, d_b = c.alloc(total), c.alloc(total)
            try:
                c.set_option("fwd_tiles", 0)
                c.warp_forward_piecewise_batch_device(np.concatenate(frames), int(ms[2]), int(ms[3]), geoms, offs, d_a)
                c.sync()
                c.set_option("fwd_tiles", -1)
                redone0 = c.redone_frames()
                c.warp_forward_piecewise_batch_device(np.concatenate(frames), int(ms[2]), int(ms[3]), geoms, offs, d_b)
                c.sync()
                # (the policy takes the tile kernels up to ~8 triangles per output tile: both meshes)
                assert c.last_forward_kernel() == 2 and c.redone_frames() == redone0, (gx, c.last_forward_kernel(), c.redone_frames() - redone0)
                if gx != 10:
                    c.set_option("fwd_tiles", 1)
                    c.warp_forward_piecewise_batch_device(np.concatenate(frames), int(ms[2]), int(ms[3]), geoms, offs, d_b)
                    c.sync()
                    assert c.last_forward_kernel() == 2 and c.redone_frames() == redone0
                for f, g in enumerate(geoms):
                    nb = g[2] * g[3] * 4
                    assert np.array_equal(c.to_host(d_a, nb, offs[f]), c.to_host(d_b, nb, offs[f])), (gx, f)
                if gx == 10:
                    g = geoms[0]
                    assert np.array_equal(c.to_host(d_b, g[2] * g[3] * 4, offs[0]).reshape(g[3], g[2], 4), _fwd_pw_oracle(sp, frames[0], tris, img, g))
            finally:
                c.free(d_a)
                c.free(d_b)
    finally:
        c.close()


def test_forward_piecewise_flagged_batches_are_redone_from_their_own_frame_sets():
    """Queued tile-binned forward batches whose frames the device flagged (tile lists over capacity) are redone by hg_sync from the
    frame set each of them was given (staged copy), not from the context's current arrays: three batches with different points
    queued with no hg_sync in between, then an inverse batch; batches into ONE buffer are redone in call order (different windows) or only
    the last of them (the same window: the later frame supersedes the earlier one)."""
    rng = np.random.default_rng(99)
    W = H = 512                                                  # 12 800 triangles: ~200 entries per tile against a first capacity of 64 -> FWD_OVERFLOW
    img = G.lcg_image(W, H, 4242)
    sp, tris = WL.grid_points(W, H, 80, 80), WL.grid_triangles(80, 80)
    ms = O.minmax_xy(sp)
    sets = []
    for k in range(3):
        dp = (sp.reshape(-1, 2) * rng.uniform(0.8, 1.0, 2) + rng.uniform(-1.5, 1.5, (sp.size // 2, 2)) + 7 + 11 * k).astype(np.float32).ravel()
        md = O.minmax_xy(dp)
        sets.append((dp, (int(md[0]), int(md[1]), int(md[2] - md[0]), int(md[3] - md[1]))))
    want = [_fwd_pw_oracle(sp, dp, tris, img, g) for dp, g in sets]
    for tail in ("forward", "inverse", "reuse", "reuse_same"):
        c = HG.Context(0)
        try:
            c.set_image(img)
            c.piecewise_set_mesh(sp, tris, int(ms[0]), int(ms[1]))
            c.set_option("fwd_tiles", 1)
            nbytes = [g[2] * g[3] * 4 for _, g in sets]
            bufs = [c.alloc(max(nbytes)) for _ in sets]
            try:
                r0 = c.redone_frames()
                for k, (dp, g) in enumerate(sets):
                    d = bufs[0] if tail.startswith("reuse") else bufs[k]
                    if tail == "reuse_same":
                        dp, g = sets[2]                          # the same window three times: only the LAST call's frame is redone
                    c.warp_forward_piecewise_batch_device(dp, int(ms[2]), int(ms[3]), [g], [0], d)
                    assert c.last_forward_kernel() == 2
                if tail == "inverse":
                    c.piecewise_set_frames(sets[0][0], [sets[0][1]], [0])
                    d_inv = c.alloc(nbytes[0])
                    c.warp_inverse_piecewise_frames_device(d_inv)
                    c.free(d_inv)
                c.sync()
                # (different windows into one buffer: every flagged frame is redone, in call order -- the last one on top)
                assert c.redone_frames() - r0 == (1 if tail == "reuse_same" else 3), (tail, c.redone_frames() - r0)
                for k, (dp, g) in enumerate(sets):
                    if tail.startswith("reuse") and k != 2:
                        continue
                    d = bufs[0] if tail.startswith("reuse") else bufs[k]
                    assert np.array_equal(c.to_host(d, nbytes[k]).reshape(g[3], g[2], 4), want[k]), (tail, k)
                # capacity grew at the sync: the next batch fits or flags again, and is right either way
                dp, g = sets[1]
                c.warp_forward_piecewise_batch_device(dp, int(ms[2]), int(ms[3]), [g], [0], bufs[1])
                c.sync()
                assert np.array_equal(c.to_host(bufs[1], nbytes[1]).reshape(g[3], g[2], 4), want[1]), tail
            finally:
                for d in bufs:
                    c.free(d)
        finally:
            c.close()


def test_sequence_fuzz_of_the_queueing_logic():
    """tools/fuzz_seq.py: random sequences of inverse / forward piecewise batches, affine frames, option changes and syncs into a small
    pool of reused output buffers, on meshes that make the device flag frames; after every sync each buffer == what the CPU oracle
    gives for the last call that wrote each byte (deferred redos in call order, staged sets, status rings).  The first version of
    this tool found a skipped redo when a later, SMALLER batch reused the buffer of a flagged one."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_seq.py"), "32", "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatching buffers" in r.stdout and " 0 frames redone" not in r.stdout, r.stdout[-500:]


def test_forward_scatter_batches_stay_on_the_device(ctx):
    """The forward (scatter) paths as asynchronous batches into GPU memory: F frames back to back == the synchronous
    single-frame calls == the oracle's sequential loops (last writer in raster order)."""
    rng = np.random.default_rng(77)
    W, H, nx, ny, F = 96, 64, 4, 3, 5
    img = G.lcg_image(W, H, 41)
    ctx.set_image(img)
    # affine frames with windows of different sizes
    mats, geoms = [], []
    for f in range(F):
        m6 = np.array([0.8 + 0.05 * f, 0.1 * f - 0.2, -0.15, 0.9, 3.0 * f, 2.0], np.float32).astype(np.float64)
        mats.append(np.concatenate([m6, [0, 0]]))
        geoms.append(tuple(int(v) for v in O.transform_limits(0, m6, W, H)))
    offs, total = HG.pack_offsets(geoms)
    d_out = ctx.alloc(max(total, 256))
    try:
        ctx.warp_forward_geometric_batch_device(0, np.concatenate(mats), geoms, offs, d_out)
        ctx.sync()
        for f in range(F):
            g = geoms[f]
            got = ctx.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
            assert np.array_equal(got, O.warp_forward_geometric(0, mats[f][:6], img, *g)), ("affine", f)
            assert np.array_equal(got, ctx.warp_forward_geometric(0, mats[f][:6], g)), ("affine sync", f)
    finally:
        ctx.free(d_out)
    # piecewise frames on one mesh
    sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
    ms = O.minmax_xy(sp)
    frames = [(sp.reshape(-1, 2) * np.float32(0.9 - 0.05 * f) + rng.uniform(-1.5, 1.5, (sp.size // 2, 2)).astype(np.float32)).astype(np.float32).ravel() for f in range(F)]
    geoms = [WL.piecewise_geom(d) for d in frames]
    ctx.piecewise_set_mesh(sp, tris, int(ms[0]), int(ms[1]))
    offs, total = HG.pack_offsets(geoms)
    d_out = ctx.alloc(max(total, 256))
    try:
        ctx.warp_forward_piecewise_batch_device(np.concatenate(frames), int(ms[2]), int(ms[3]), geoms, offs, d_out)
        ctx.sync()
        fmap = O.build_tri_map(sp, tris, int(ms[2] - ms[0]), int(ms[1]), int((ms[2] - ms[0]) * (ms[3] - ms[1])))
        for f in range(F):
            g = geoms[f]
            got = ctx.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
            fwd = O.piecewise_matrices(sp, frames[f], tris)
            want = O.warp_forward_piecewise(fmap, fwd, img, int(ms[0]), int(ms[1]), int(ms[2]), int(ms[3]), *g)
            assert np.array_equal(got, want), ("piecewise", f)
            assert np.array_equal(got, ctx.warp_forward_piecewise(frames[f], int(ms[2]), int(ms[3]), g)), ("piecewise sync", f)
    finally:
        ctx.free(d_out)


def test_forward_warps_with_one_source_per_frame():
    """The video loop `for (f) warp(frame_f)` when warp() dispatches FORWARD (Homography.js:421, :426, README.md:121-137): frame f of a
    forward batch reads image f mod n_images (hg_set_images_device).  Affine and piecewise, scatter + gather and the tile kernels,
    each frame against the oracle's sequential loop on ITS image."""
    rng = np.random.default_rng(4242)
    W, H, nx, ny, F, NI = 320, 256, 5, 4, 7, 3
    imgs = [G.lcg_image(W, H, 500 + k) for k in range(NI)]
    stride = W * H * 4 + 128
    c = HG.Context(0)
    d_src = c.alloc(stride * NI)
    try:
        for k in range(NI):
            c.to_device(d_src, imgs[k], k * stride)
        c.set_images_device(d_src, W, H, NI, stride)
        # affine frames (same-size windows are what warp() sends forward, :426; any window is legal at the C ABI)
        mats, geoms = [], []
        for f in range(F):
            m6 = np.array([0.9 + 0.03 * f, 0.05 * f - 0.1, -0.08, 0.95, 2.0 * f, 1.0], np.float32).astype(np.float64)
            mats.append(np.concatenate([m6, [0, 0]]))
            geoms.append(tuple(int(v) for v in O.transform_limits(0, m6, W, H)))
        offs, total = HG.pack_offsets(geoms)
        d_out = c.alloc(max(total, 256))
        try:
            for tiles in (0, 1):
                c.set_option("fwd_tiles", tiles)
                c.warp_forward_geometric_batch_device(0, np.concatenate(mats), geoms, offs, d_out)
                c.sync()
                assert c.last_forward_kernel() == 1 + tiles
                for f in range(F):
                    g = geoms[f]
                    got = c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
                    assert np.array_equal(got, O.warp_forward_geometric(0, mats[f][:6], imgs[f % NI], *g)), ("affine", tiles, f)
        finally:
            c.free(d_out)
        # piecewise frames on one mesh (shrunk so that warp() would dispatch forward, :421)
        sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
        ms = O.minmax_xy(sp)
        frames = [(sp.reshape(-1, 2) * np.float32(0.95 - 0.01 * f) + rng.uniform(-2.0, 2.0, (sp.size // 2, 2)).astype(np.float32)).astype(np.float32).ravel() for f in range(F)]
        geoms = [WL.piecewise_geom(d) for d in frames]
        c.piecewise_set_mesh(sp, tris, int(ms[0]), int(ms[1]))
        offs, total = HG.pack_offsets(geoms)
        d_out = c.alloc(max(total, 256))
        try:
            fmap = O.build_tri_map(sp, tris, int(ms[2] - ms[0]), int(ms[1]), int((ms[2] - ms[0]) * (ms[3] - ms[1])))
            for tiles in (0, 1):
                c.set_option("fwd_tiles", tiles)
                c.warp_forward_piecewise_batch_device(np.concatenate(frames), int(ms[2]), int(ms[3]), geoms, offs, d_out)
                c.sync()
                assert c.last_forward_kernel() == 1 + tiles
                for f in range(F):
                    g = geoms[f]
                    got = c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
                    fwd = O.piecewise_matrices(sp, frames[f], tris)
                    want = O.warp_forward_piecewise(fmap, fwd, imgs[f % NI], int(ms[0]), int(ms[1]), int(ms[2]), int(ms[3]), *g)
                    assert np.array_equal(got, want), ("piecewise", tiles, f)
        finally:
            c.free(d_out)
    finally:
        c.set_image(imgs[0])                                 # drop the alias before the buffer goes away
        c.free(d_src)
        c.close()


def test_batch_frames_equal_single_frames(ctx):
    """F destination point sets in one launch == F single-frame calls (frames differ in geometry and offsets)."""
    W, H, nx, ny, F = 320, 200, 10, 6, 5
    img = G.lcg_image(W, H, 77)
    sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
    frames = [WL.sin_dst(sp, 6.0 + f, 8 + (f % 4)) for f in range(F)]
    geoms = [WL.piecewise_geom(d) for d in frames]
    mm = O.minmax_xy(sp)
    ctx.set_image(img)
    ctx.piecewise_set_mesh(sp, tris, int(mm[0]), int(mm[1]))
    offs, total = HG.pack_offsets(geoms)
    d_out = ctx.alloc(total)
    try:
        ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
        ctx.warp_inverse_piecewise_frames_device(d_out)
        ctx.sync()
        for f in range(F):
            g = geoms[f]
            got = ctx.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
            want = O.warp_inverse_piecewise(sp, frames[f], tris, img, int(mm[0]), int(mm[1]), *g)
            assert np.array_equal(got, want), f
    finally:
        ctx.free(d_out)
    # geometric batch
    mats, ggeoms = [], []
    s4 = np.array([0, 0, 0, H, W, 0, W, H], np.float32)
    for f in range(F):
        d4 = WL.projective_dst(W, H, 0.02 * f)
        fw = O.projective_from_squares(s4, d4)
        ggeoms.append(tuple(int(v) for v in O.transform_limits(1, fw, W, H)))
        mats.append(HG.solve_projective(d4, s4))
    offs, total = HG.pack_offsets(ggeoms)
    d_out = ctx.alloc(total)
    try:
        ctx.geometric_set_frames(1, np.concatenate(mats), ggeoms, offs)
        ctx.warp_inverse_geometric_frames_device(d_out)
        ctx.sync()
        for f in range(F):
            g = ggeoms[f]
            got = ctx.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
            assert np.array_equal(got, O.warp_inverse_geometric(1, mats[f], img, *g)), f
    finally:
        ctx.free(d_out)


@pytest.mark.parametrize("name", ["C3_batch_4k", "C5_batch_8k", "C4_orbit_4k"])
def test_full_size_batches_match_reference_shas(ctx, name):
    """The benchmarked caller loop at FULL size as ONE launch: C3 / C5 with sin((8..11) x / pi) (F = 4) and eight frames of the
    C4 face-mesh orbit (F = 8); every frame's RGBA SHA-256 and hit count against what the reference itself produced."""
    case = [c for c in GOLD["cases"] if c["name"] == name][0]
    ws = case["warps"]
    img = G.case_images(case)["a"]
    sp, tris = G.f32_from_bits(ws[0]["srcPoints"]), G.case_triangles(case)
    frames = [G.f32_from_bits(w["dstPoints"]) for w in ws]
    geoms = [(w["xOff"], w["yOff"], w["objW"], w["objH"]) for w in ws]
    assert all(w["path"] == "_inversePiecewiseAffineWarp" for w in ws)
    ctx.set_image(img)
    ctx.piecewise_set_mesh(sp, tris, ws[0]["minSrcX"], ws[0]["minSrcY"])
    offs, total = HG.pack_offsets(geoms)
    d_out = ctx.alloc(total)
    try:
        ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
        ctx.warp_inverse_piecewise_frames_device(d_out)
        ctx.sync()
        for f, w in enumerate(ws):
            got = ctx.to_host(d_out, w["objW"] * w["objH"] * 4, offs[f])
            assert G.sha256(got) == w["out"]["sha"], (name, f)
        # N_hit of every frame on an all-255 source (one more launch): the algorithmic read bytes bench.py prices
        ctx.set_image(np.full_like(img, 255))
        ctx.warp_inverse_piecewise_frames_device(d_out)
        ctx.sync()
        for f, w in enumerate(ws):
            got = ctx.to_host(d_out, w["objW"] * w["objH"] * 4, offs[f]).reshape(-1, 4)
            assert int((got[:, 3] == 255).sum()) == w["nhit"], (name, f)
    finally:
        ctx.free(d_out)


def test_one_source_per_frame(ctx):
    """hg_set_images_device: frame f of a frame set reads image f % n_images (the video case, every warp() its own image).
    Piecewise (incl. a frame forced through the map path) and projective, against the oracle frame by frame."""
    W, H, nx, ny, F, NI = 288, 180, 9, 6, 7, 3
    imgs = [G.lcg_image(W, H, 100 + k) for k in range(NI)]
    stride = W * H * 4 + 64                                 # images need not be densely packed
    sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
    frames = [WL.sin_dst(sp, 5.0 + f, 8 + (f % 4)) for f in range(F)]
    geoms = [WL.piecewise_geom(d) for d in frames]
    mm = O.minmax_xy(sp)
    d_src = ctx.alloc(stride * NI)
    offs, total = HG.pack_offsets(geoms)
    d_out = ctx.alloc(total)
    try:
        for k in range(NI):
            ctx.to_device(d_src, imgs[k], k * stride)
        ctx.set_images_device(d_src, W, H, NI, stride)
        ctx.piecewise_set_mesh(sp, tris, int(mm[0]), int(mm[1]))
        ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
        ctx.warp_inverse_piecewise_frames_device(d_out)
        ctx.sync()
        for f in range(F):
            g = geoms[f]
            got = ctx.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
            want = O.warp_inverse_piecewise(sp, frames[f], tris, imgs[f % NI], int(mm[0]), int(mm[1]), *g)
            assert np.array_equal(got, want), ("piecewise", f)
        mats, ggeoms = [], []
        s4 = np.array([0, 0, 0, H, W, 0, W, H], np.float32)
        for f in range(F):
            d4 = WL.projective_dst(W, H, 0.02 * f)
            ggeoms.append(tuple(int(v) for v in O.transform_limits(1, O.projective_from_squares(s4, d4), W, H)))
            mats.append(HG.solve_projective(d4, s4))
        goffs, gtotal = HG.pack_offsets(ggeoms)
        assert gtotal <= total
        ctx.geometric_set_frames(1, np.concatenate(mats), ggeoms, goffs)
        ctx.warp_inverse_geometric_frames_device(d_out)
        ctx.sync()
        for f in range(F):
            g = ggeoms[f]
            got = ctx.to_host(d_out, g[2] * g[3] * 4, goffs[f]).reshape(g[3], g[2], 4)
            assert np.array_equal(got, O.warp_inverse_geometric(1, mats[f], imgs[f % NI], *g)), ("projective", f)
    finally:
        ctx.set_image(imgs[0])                               # drop the alias before the buffer goes away
        ctx.free(d_out)
        ctx.free(d_src)


def test_one_fma_predicate_boundary_on_random_meshes(ctx):
    """Random small meshes straddling hg_affine_one_fma_form's boundary: translations with vertices moved by 1-8 units in the last place
    (shears of 2^-27 ... 2^-17 against the unit scale: exact sums on one side of the predicate, inexact on the other), random
    stretches on top; 8 frames per set, the self-span kernel forced.  Every frame against the oracle; both kinds of frame must occur."""
    rng = np.random.default_rng(20260929)
    W, H = 96, 80
    img = G.lcg_image(W, H, 5)
    sp, tris = WL.grid_points(W, H, 4, 3), WL.grid_triangles(4, 3)
    ms = WL.src_min(sp)
    seen = set()
    with HG.Context(0) as c:
        c.set_image(img); c.piecewise_set_mesh(sp, tris, ms[0], ms[1]); c.set_option("min_row_groups", 0)
        for trial in range(30):
            frames = []
            for f in range(8):
                d = (sp.reshape(-1, 2) + rng.integers(0, 6, 2).astype(np.float32)).astype(np.float32)
                if rng.random() < 0.3: d[:, 0] = (d[:, 0] * np.float32(rng.choice([1.25, 0.75, 1.5]))).astype(np.float32)
                for _ in range(int(rng.integers(0, 4))):
                    v, ax, n = int(rng.integers(0, d.shape[0])), int(rng.integers(0, 2)), int(rng.integers(1, 9))
                    for _ in range(n): d[v, ax] = np.nextafter(d[v, ax], np.float32(1e9 if rng.random() < 0.5 else -1e9), dtype=np.float32)
                frames.append(d.ravel())
            geoms = [WL.piecewise_geom(d) for d in frames]
            for f in range(8):
                fwd = HG.solve_affine_triangles(sp, frames[f], tris).reshape(-1, 6)
                seen.add(all(HG.affine_one_fma_form(HG.invert_affine(m), geoms[f]) for m in fwd))
            offs, total = HG.pack_offsets(geoms)
            d_out = c.alloc(total)
            try:
                c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                c.warp_inverse_piecewise_frames_device(d_out); c.sync()
                for f in range(8):
                    g = geoms[f]
                    want = O.warp_inverse_piecewise(sp, frames[f], tris, img, ms[0], ms[1], *g)
                    assert np.array_equal(c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4), want), (trial, f)
            finally:
                c.free(d_out)
    assert seen == {True, False}


def test_download_stream_protocol_with_a_flagged_frame(ctx):
    """hg_download_behind_warps / hg_fence_downloads, used the way the Node addon pipelines warpBatch({images}): frame f's pixels go down a
    third stream right behind its warp, UNSETTLED, while the host binds image f + 1 (which settles frame f) and warps it.  Frame 2 has a
    NaN vertex: the fused kernel only flags it, the settlement redoes it through the map -- hg_redone_frames moves -- and the frame is
    downloaded again.  Every frame equals the oracle."""
    W, H, nx, ny, F = 288, 180, 9, 6, 5
    imgs = [G.lcg_image(W, H, 500 + k) for k in range(F)]
    sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
    frames = [WL.sin_dst(sp, 4.0 + f, 8 + (f % 4)) for f in range(F)]
    frames[2] = frames[2].copy(); frames[2][4] = np.nan            # (an x coordinate: its triangles keep their rows and are irregular)
    geoms = [WL.piecewise_geom(frames[0])] * F                           # (one window for all: the NaN frame has none of its own)
    ms = WL.src_min(sp)
    offs, total = HG.pack_offsets(geoms)
    stride = W * H * 4
    with HG.Context(0) as c:
        d_src, d_out = c.alloc(stride * F), c.alloc(total)
        outs = [np.zeros(geoms[f][2] * geoms[f][3] * 4, np.uint8) for f in range(F)]
        try:
            for k in range(F): c.to_device(d_src, imgs[k], k * stride)
            c.set_image_device(d_src, W, H); c.piecewise_set_mesh(sp, tris, ms[0], ms[1])
            c.set_option("min_row_groups", 0)
            redone = c.redone_frames()
            for f in range(F + 1):
                if f < F: c.set_image_device(d_src + f * stride, W, H)       # settles frame f - 1
                else: c.sync()
                if f > 0 and c.redone_frames() != redone:
                    c.download_behind_warps(outs[f - 1], d_out, offs[f - 1]); redone = c.redone_frames()
                if f == F: break
                c.piecewise_set_frames(frames[f], [geoms[f]], [offs[f]])
                c.warp_inverse_piecewise_frames_device(d_out)
                c.download_behind_warps(outs[f], d_out, offs[f])
            c.fence_downloads()
            assert c.redone_frames() >= 1
            for f in range(F):
                g = geoms[f]
                want = O.warp_inverse_piecewise(sp, frames[f], tris, imgs[f], ms[0], ms[1], *g)
                assert np.array_equal(outs[f].reshape(g[3], g[2], 4), want), f
        finally:
            c.set_image(imgs[0])
            c.free(d_out); c.free(d_src)


def test_frame_sets_mixing_one_fma_and_two_rounding_frames(ctx):
    """k_pw_rows<SELF> evaluates a frame's coordinates with one fma where k_tri_setup found the sums of every triangle of the frame exact
    (hg_affine_one_fma_form) and with the reference's two roundings otherwise -- per frame, decided on the device at every step.
    Frame set: pure translations and ordinary deformations (every sum exact) next to frames in which a few vertices are moved by one ulp
    (shears of 2^-23: far too small against the unit scale for exact sums).  Every frame against the oracle, with the self-span kernel
    forced (min_row_groups = 0) and under the default policy, safe spans on and off; then the same buffers again with the frames in another
    order (a frame index changes kind from one step to the next)."""
    W = H = 256
    img = G.lcg_image(W, H, 77)
    sp, tris = WL.grid_points(W, H, 4, 4), WL.grid_triangles(4, 4)
    base = (sp.reshape(-1, 2) + np.array([3, 2], np.float32)).astype(np.float32)
    frames = []
    for f in range(8):
        d = base.copy()
        if f in (1, 2, 4, 7):
            for v in ((6 + f) % 25, (12 + 2 * f) % 25, 18):                # vertices nudged by one unit in the last place
                d[v, f % 2] = np.nextafter(d[v, f % 2], np.float32(1e9), dtype=np.float32)
        if f in (3, 4): d[7] += np.float32(0.375)                           # + an ordinary deformation
        if f == 5: d[:, 0] = (d[:, 0] * np.float32(1.25)).astype(np.float32)
        frames.append(d.ravel())
    geoms = [WL.piecewise_geom(d) for d in frames]
    ms = WL.src_min(sp)
    kinds = []
    for f in range(8):
        fwd = HG.solve_affine_triangles(sp, frames[f], tris).reshape(-1, 6)
        kinds.append(all(HG.affine_one_fma_form(HG.invert_affine(m), geoms[f]) for m in fwd))
    assert kinds == [True, False, False, True, False, True, True, False], kinds
    with HG.Context(0) as c:
        c.set_image(img); c.piecewise_set_mesh(sp, tris, ms[0], ms[1])
        for order in (list(range(8)), [1, 0, 3, 2, 7, 6, 5, 4]):
            fr = [frames[i] for i in order]; gm = [geoms[i] for i in order]
            offs, total = HG.pack_offsets(gm)
            want = [O.warp_inverse_piecewise(sp, fr[f], tris, img, ms[0], ms[1], *gm[f]) for f in range(8)]
            d_out = c.alloc(total)
            try:
                for mrg in (0, -1):
                    for safe in (1, 0, -1):
                        if mrg >= 0: c.set_option("min_row_groups", mrg)
                        c.set_option("safe_spans", safe)
                        c.piecewise_set_frames(np.concatenate(fr), gm, offs)
                        for step in range(2):                               # (resident points: the second step re-derives the per-frame choice)
                            c.warp_inverse_piecewise_frames_device(d_out); c.sync()
                            for f in range(8):
                                g = gm[f]
                                assert np.array_equal(c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4), want[f]), (order, mrg, safe, step, f)
                        if mrg == 0: assert c.last_piecewise_self() != 0
            finally:
                c.free(d_out)


def test_geometric_frame_sets_of_uneven_frames_on_their_own_sources(ctx):
    """k_geo_fast's XCD bands rotate with the frame when every frame reads its own source: frame sets whose frames differ in size by two
    orders of magnitude (rows past a frame's end, bands a small frame does not reach), affine and projective, with each rotation setting."""
    W, H, NI = 320, 200, 4
    imgs = [G.lcg_image(W, H, 300 + k) for k in range(NI)]
    stride = W * H * 4 + 256
    rng = np.random.default_rng(5)
    d_src = ctx.alloc(stride * NI)
    try:
        for k in range(NI):
            ctx.to_device(d_src, imgs[k], k * stride)
        ctx.set_images_device(d_src, W, H, NI, stride)
        for kind in (0, 1):
            mats, geoms = [], []
            for f in range(11):
                sc = [1.0, 0.04, 0.6, 2.3, 0.015, 1.3, 0.3, 3.1, 0.9, 0.11, 1.7][f]
                if kind == 0:
                    s = np.array([0, 0, 0, H, W, 0], np.float32)
                    d = (np.array([3, 2, 10, H, W, 7], np.float32) * np.float32(sc) + rng.uniform(-2, 2, 6).astype(np.float32)).astype(np.float32)
                    fwd = HG.solve_affine(s, d).astype(np.float64); inv = HG.solve_affine(d, s).astype(np.float64)
                    lim = HG.transform_limits(0, fwd, W, H)
                    mats.append(np.concatenate([inv, [0.0, 0.0]]))
                else:
                    s = WL.corners(W, H)
                    d = (WL.projective_dst(W, H, 0.01 * f) * np.float32(sc)).astype(np.float32)
                    lim = HG.transform_limits(1, HG.solve_projective(s, d), W, H)
                    mats.append(HG.solve_projective(d, s))
                geoms.append(tuple(int(v) for v in lim))
            offs, total = HG.pack_offsets(geoms)
            d_out = ctx.alloc(total)
            try:
                want = [O.warp_inverse_geometric(kind, mats[f][:6] if kind == 0 else mats[f], imgs[f % NI], *geoms[f]) for f in range(11)]
                for rot in (-1, 0, 1):
                    ctx.set_option("xcc_rotate", rot)
                    ctx.geometric_set_frames(kind, np.concatenate(mats), geoms, offs)
                    ctx.warp_inverse_geometric_frames_device(d_out)
                    ctx.sync()
                    for f in range(11):
                        g = geoms[f]
                        got = ctx.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
                        assert np.array_equal(got, want[f]), (kind, rot, f, g)
            finally:
                ctx.set_option("xcc_rotate", -1)
                ctx.free(d_out)
    finally:
        ctx.set_image(imgs[0])
        ctx.free(d_src)


def test_device_side_frame_solves(ctx):
    """hg_geometric_set_frames_points: the inverse matrices are solved on the device at every warp (k_solve_frames, one lane
    per frame, numeric.js LU order).  Bit patterns == host solve == golden vectors; the warped frames == frames warped from
    host-solved matrices == oracle; incl. a frame whose window crosses the horizon (IEEE divides) next to plain-range ones."""
    f = GOLD["func"]
    srcs = np.stack([G.f32_from_bits(v["src"]) for v in f["projective"]])
    dsts = np.stack([G.f32_from_bits(v["dst"]) for v in f["projective"]])
    want = np.stack([G.f64_from_hex(v["out"]) for v in f["projective"]])
    n = len(srcs)
    ctx.set_image(G.lcg_image(64, 48, 9))
    ctx.geometric_set_frames_points(1, srcs.ravel(), dsts.ravel(), [(0, 0, 8, 8)] * n)
    got = ctx.get_geometric_matrices(n)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)) or all(_nan_eq64(a, b) for a, b in zip(got, want))
    # random squares: device == host, bit for bit
    rng = np.random.default_rng(99)
    S = rng.uniform(-500, 3000, (257, 8)).astype(np.float32); D = rng.uniform(-500, 3000, (257, 8)).astype(np.float32)
    S[5] = S[6]; D[7, 2:4] = D[7, 0:2]                       # + a degenerate one
    ctx.geometric_set_frames_points(1, S.ravel(), D.ravel(), [(0, 0, 4, 4)] * 257)
    got = ctx.get_geometric_matrices(257)
    host = np.stack([HG.solve_projective(S[k], D[k]) for k in range(257)])
    assert all(_nan_eq64(a, b) for a, b in zip(got, host))
    A3s = rng.uniform(-100, 900, (65, 6)).astype(np.float32); A3d = rng.uniform(-100, 900, (65, 6)).astype(np.float32)
    ctx.geometric_set_frames_points(0, A3s.ravel(), A3d.ravel(), [(0, 0, 4, 4)] * 65)
    got = ctx.get_geometric_matrices(65)[:, :6]
    host = np.stack([HG.solve_affine(A3s[k], A3d[k]).astype(np.float64) for k in range(65)])
    assert all(_nan_eq64(a, b) for a, b in zip(got, host))
    # warps: points path == matrices path == oracle
    W, H, F = 320, 200, 6
    img = G.lcg_image(W, H, 31)
    ctx.set_image(img)
    s4 = WL.corners(W, H)
    d4s = [WL.projective_dst(W, H, 0.03 * k) for k in range(F - 1)]
    d4s.append(np.array([40, 10, 10, 190, 300, 60, 310, 120], np.float32))
    geoms = [tuple(int(v) for v in O.transform_limits(1, O.projective_from_squares(s4, d4), W, H)) for d4 in d4s]
    geoms[-1] = (-400, -300, 1200, 900)                      # a window far larger than the quad: the horizon crosses it
    mats = [HG.solve_projective(d4, s4) for d4 in d4s]
    assert not HG.projective_plain_range(mats[-1], geoms[-1]) and HG.projective_plain_range(mats[0], geoms[0])
    offs, total = HG.pack_offsets(geoms)
    d_out = ctx.alloc(total)
    try:
        ctx.geometric_set_frames_points(1, np.concatenate(d4s), np.tile(s4, F), geoms, offs)
        ctx.warp_inverse_geometric_frames_device(d_out)
        ctx.sync()
        for k in range(F):
            g = geoms[k]
            got = ctx.to_host(d_out, g[2] * g[3] * 4, offs[k]).reshape(g[3], g[2], 4)
            assert np.array_equal(got, O.warp_inverse_geometric(1, mats[k], img, *g)), k
        a3s = np.array([0, 0, 0, H, W, 0], np.float32)
        a3d = [WL.affine_dst(W, H, 0.01 * k) for k in range(F)]
        ageoms = [tuple(int(v) for v in O.transform_limits(0, O.affine_from_triangles(a3s, d).astype(np.float64), W, H)) for d in a3d]
        aoffs, atotal = HG.pack_offsets(ageoms)
        assert atotal <= total
        ctx.geometric_set_frames_points(0, np.concatenate(a3d), np.tile(a3s, F), ageoms, aoffs)
        ctx.warp_inverse_geometric_frames_device(d_out)
        ctx.sync()
        for k in range(F):
            g = ageoms[k]
            got = ctx.to_host(d_out, g[2] * g[3] * 4, aoffs[k]).reshape(g[3], g[2], 4)
            m = HG.solve_affine(a3d[k], a3s).astype(np.float64)
            assert np.array_equal(got, O.warp_inverse_geometric(0, m, img, *g)), ("affine", k)
    finally:
        ctx.free(d_out)


def _nan_eq64(a, b):
    a, b = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64)
    return bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


def test_failed_set_frames_leaves_no_frame_set(ctx):
    """A frame set that fails validation part-way (offset beyond 2^26 in the LAST frame) must not leave the new
    host-side frames over device buffers sized for the old set: the next warp reports HG_ERR_STATE."""
    W, H = 64, 48
    ctx.set_image(G.lcg_image(W, H, 5))
    sp, tris = WL.grid_points(W, H, 2, 2), WL.grid_triangles(2, 2)
    ctx.piecewise_set_mesh(sp, tris, 0, 0)
    dp = WL.sin_dst(sp, 2.0, 8)
    g = WL.piecewise_geom(dp)
    ctx.piecewise_set_frames(dp, [g])
    d_out = ctx.alloc(g[2] * g[3] * 4 * 4)
    try:
        ctx.warp_inverse_piecewise_frames_device(d_out)
        ctx.sync()
        bad = [g, g, g, (1 << 27, 0, g[2], g[3])]
        with pytest.raises(HG.HgError):
            ctx.piecewise_set_frames(np.concatenate([dp] * 4), bad)
        with pytest.raises(HG.HgError) as e:
            ctx.warp_inverse_piecewise_frames_device(d_out)
        assert e.value.code == 4                             # HG_ERR_STATE
        m = HG.solve_projective(WL.projective_dst(W, H), WL.corners(W, H))
        ctx.geometric_set_frames(1, m, [(0, 0, W, H)])
        with pytest.raises(HG.HgError):
            ctx.geometric_set_frames(1, np.concatenate([m] * 2), [(0, 0, W, H), (0, 1 << 27, W, H)])
        with pytest.raises(HG.HgError) as e:
            ctx.warp_inverse_geometric_frames_device(d_out)
        assert e.value.code == 4
    finally:
        ctx.free(d_out)


def test_irregular_and_overflow_frames_fall_back_to_map_path(ctx):
    """Spans wider than the whole map, and rows crossed by more spans than the row lists hold (the frame is then redone
    by the library through the materialised map): both still match the oracle bit for bit."""
    W, H = 64, 48
    img = G.lcg_image(W, H, 9)
    # (a) output map of 3 rows, triangles wider than len/…: tiny obj_h with x-extent >= len is impossible to reach
    #     through the reference geometry, so drive the C ABI directly with an explicit (inconsistent) window.
    sp = np.array([0, 0, 63, 0, 0, 47, 63, 47], np.float32)
    dp = np.array([0, 0, 300, 0, 0, 2, 300, 2], np.float32)
    tris = np.array([0, 1, 2, 1, 3, 2], np.uint32)
    geom = (0, 0, 20, 3)          # len = 60 < triangle width 300: spans run over several output rows and past the map end
    ctx.set_image(img)
    ctx.piecewise_set_mesh(sp, tris, 0, 0)
    ctx.piecewise_prepare(dp, geom)
    want, wmap, _, _ = O.warp_inverse_piecewise(sp, dp, tris, img, 0, 0, *geom, taps=True)
    assert np.array_equal(ctx.warp_inverse_piecewise(), want)
    assert np.array_equal(ctx.get_tri_map(fused=True), wmap)
    assert np.array_equal(ctx.get_tri_map(), wmap)
    # (b) 1100 thin triangles crossing every row -> more spans per row than the lists hold -> FRAME_LDS_OVERFLOW -> map path
    n = 1100
    W2, H2 = 2400, 8
    img2 = G.lcg_image(W2, H2, 10)
    xs = np.linspace(0, W2, n + 1)
    sp2 = np.stack([np.repeat(xs, 2), np.tile([0.0, H2], n + 1)], 1).astype(np.float32).ravel()
    tr2 = np.array([[2 * i, 2 * i + 2, 2 * i + 1] for i in range(n)], np.uint32).ravel()
    dp2 = sp2.copy()
    dp2[1::2] *= 1.5
    mm, md = O.minmax_xy(sp2), O.minmax_xy(dp2)
    g2 = (int(md[0]), int(md[1]), int(md[2] - md[0]), int(md[3] - md[1]))
    ctx.set_image(img2)
    ctx.piecewise_set_mesh(sp2, tr2, int(mm[0]), int(mm[1]))
    ctx.piecewise_prepare(dp2, g2)
    want2 = O.warp_inverse_piecewise(sp2, dp2, tr2, img2, int(mm[0]), int(mm[1]), *g2)
    assert np.array_equal(ctx.warp_inverse_piecewise(), want2)
    assert np.array_equal(ctx.warp_inverse_piecewise_via_map(), want2)


def test_tile_kernel_policy_and_fallback():
    """k_pw_tile (8-row x 2048-column tiles, gathers along the source rows): (a) the default exactly when every frame reads its own source,
    never with a shared one unless forced, same bytes as k_pw_patch and the oracle either way, windows that wrap over the row end and
    frames narrower than a tile included; (b) a row with more spans inside one tile than its LDS block holds (96) flags the frame: redone
    through the map path, bit-exact, and the mesh goes back to k_pw_patch."""
    c = HG.Context(0)
    try:
        W, H, nx, ny, F, NI = 2600, 520, 12, 6, 5, 2
        imgs = [G.lcg_image(W, H, 700 + k) for k in range(NI)]
        sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
        frames = [WL.sin_dst(sp, 30.0 + 9 * f, 8 + (f % 4)) for f in range(F)]     # steep shear: runs change rows inside a block
        geoms = [WL.piecewise_geom(d) for d in frames]
        ms = WL.src_min(sp)
        offs, total = HG.pack_offsets(geoms)
        stride = W * H * 4
        d_src, d_out = c.alloc(stride * NI), c.alloc(total)
        try:
            for k in range(NI):
                c.to_device(d_src, imgs[k], k * stride)
            c.set_option("min_row_groups", 0)
            c.piecewise_set_mesh(sp, tris, ms[0], ms[1])
            wants = {n: [O.warp_inverse_piecewise(sp, frames[f], tris, imgs[f % n], ms[0], ms[1], *geoms[f]) for f in range(F)] for n in (1, NI)}
            for n_img, tile, kernel in ((NI, -1, 5), (NI, 0, 3), (1, -1, 3), (1, 1, 5)):
                c.set_images_device(d_src, W, H, n_img, stride)
                c.set_option("patch", 1); c.set_option("self_spans", 1); c.set_option("tile", tile)
                c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                c.warp_inverse_piecewise_frames_device(d_out)
                c.sync()
                assert c.last_piecewise_kernel() == kernel, (n_img, tile, c.last_piecewise_kernel())
                for f in range(F):
                    g = geoms[f]
                    assert np.array_equal(c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4), wants[n_img][f]), (n_img, tile, f)
            assert c.redone_frames() == 0
        finally:
            c.free(d_out)
        # (b) 70 x 2 cells on 2000 columns: ~140 spans per row, all inside ONE tile (k_pw_patch holds 199 per row)
        W2, H2, nx, ny, F = 2000, 160, 70, 2, 3
        img = G.lcg_image(W2, H2, 811)
        sp, tris = WL.grid_points(W2, H2, nx, ny), WL.grid_triangles(nx, ny)
        frames = [WL.sin_dst(sp, 3.0 + f, 8 + f) for f in range(F)]
        geoms = [WL.piecewise_geom(d) for d in frames]
        ms = WL.src_min(sp)
        offs, total = HG.pack_offsets(geoms)
        d_out = c.alloc(total)
        try:
            c.set_image(img)
            c.piecewise_set_mesh(sp, tris, ms[0], ms[1])
            c.set_option("tile", 1)
            want = [O.warp_inverse_piecewise(sp, frames[f], tris, img, ms[0], ms[1], *geoms[f]) for f in range(F)]
            kernels = []
            for rep in range(2):
                c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                c.warp_inverse_piecewise_frames_device(d_out)
                c.sync()
                kernels.append(c.last_piecewise_kernel())
                for f in range(F):
                    g = geoms[f]
                    assert np.array_equal(c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4), want[f]), ("overflow", rep, f)
            assert kernels == [5, 3] and c.redone_frames() == F, (kernels, c.redone_frames())
        finally:
            c.free(d_out)
            c.free(d_src)
    finally:
        c.close()


def test_self_span_path_policy_flags_and_falls_back():
    """The self-span path (k_tri_setup + k_pw_rows<SELF> / k_pw_patch<SELF>: the row workgroups evaluate the spans of their own rows, no
    row lists): (a) taken by default for a frame set that fills the chip, not for a single frame, forced / forbidden by the option, same
    bytes either way and equal to the oracle; (b) k_pw_patch<SELF> with candidate bands (a mesh beyond 256 triangles, one source per
    frame) == row lists == oracle; (c) a mesh with more spans per row than its LDS blocks hold flags its frames: they are redone through
    the map path (still bit-exact) and the context returns to row lists for that mesh."""
    c = HG.Context(0)
    try:
        # (a) policy
        W, H, nx, ny, F = 1024, 768, 8, 6, 12
        img = G.lcg_image(W, H, 61)
        sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
        frames = [WL.sin_dst(sp, 6.0 + f, 8 + (f % 4)) for f in range(F)]
        geoms = [WL.piecewise_geom(d) for d in frames]
        ms = WL.src_min(sp)
        offs, total = HG.pack_offsets(geoms)
        c.set_image(img)
        c.piecewise_set_mesh(sp, tris, ms[0], ms[1])
        d_out = c.alloc(total)
        try:
            want = [O.warp_inverse_piecewise(sp, frames[f], tris, img, ms[0], ms[1], *geoms[f]) for f in range(F)]
            seen = {}
            for opt in (-1, 0, 1):
                c.set_option("self_spans", opt)
                c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                c.warp_inverse_piecewise_frames_device(d_out)
                c.sync()
                seen[opt] = c.last_piecewise_self()
                for f in range(F):
                    g = geoms[f]
                    assert np.array_equal(c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4), want[f]), (opt, f)
            assert seen == {-1: 1, 0: 0, 1: 1}, seen               # 12 frames x 192 four-row groups = 2304 >= min_row_groups: self-spans by default
            c.set_option("self_spans", -1)
            c.piecewise_prepare(frames[0], geoms[0])                 # a single frame: small set -> row lists by default
            assert np.array_equal(c.warp_inverse_piecewise(), want[0]) and c.last_piecewise_self() == 0
            c.set_option("self_spans", 1)
            c.piecewise_prepare(frames[0], geoms[0])                 # ... forced: the short-latency prologue instantiation
            assert np.array_equal(c.warp_inverse_piecewise(), want[0]) and c.last_piecewise_self() == 1
            assert np.array_equal(c.get_tri_map(fused=True), O.warp_inverse_piecewise(sp, frames[0], tris, img, ms[0], ms[1], *geoms[0], taps=True)[1])
            assert c.redone_frames() == 0
        finally:
            c.free(d_out)
        # (b) k_pw_patch<SELF> with bands: 30 x 20 cells = 1200 triangles, one source per frame
        W, H, nx, ny, F, NI = 1536, 1024, 30, 20, 6, 3
        imgs = [G.lcg_image(W, H, 300 + k) for k in range(NI)]
        sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
        frames = [WL.sin_dst(sp, 9.0 + f, 8 + (f % 4)) for f in range(F)]
        geoms = [WL.piecewise_geom(d) for d in frames]
        ms = WL.src_min(sp)
        offs, total = HG.pack_offsets(geoms)
        stride = W * H * 4
        d_src, d_out = c.alloc(stride * NI), c.alloc(total)
        try:
            for k in range(NI):
                c.to_device(d_src, imgs[k], k * stride)
            c.set_images_device(d_src, W, H, NI, stride)
            c.set_option("min_row_groups", 0)
            c.piecewise_set_mesh(sp, tris, ms[0], ms[1])
            want = [O.warp_inverse_piecewise(sp, frames[f], tris, imgs[f % NI], ms[0], ms[1], *geoms[f]) for f in range(F)]
            for opt, tile, kernel in ((1, 0, 3), (0, -1, 3), (1, -1, 5)):   # (one source per frame: k_pw_tile by default where spans are self-evaluated)
                c.set_option("self_spans", opt)
                c.set_option("patch", 1)
                c.set_option("tile", tile)
                c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                c.warp_inverse_piecewise_frames_device(d_out)
                c.sync()
                assert c.last_piecewise_kernel() == kernel and c.last_piecewise_self() == opt, (opt, tile, c.last_piecewise_kernel(), c.last_piecewise_self())
                for f in range(F):
                    g = geoms[f]
                    assert np.array_equal(c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4), want[f]), ("bands", opt, f)
            assert c.redone_frames() == 0
        finally:
            c.set_image(imgs[0])
            c.free(d_out)
            c.free(d_src)
        # (b2) k_pw_rows<SELF> with candidate bands: sparse rows (12 cells across), 1440 triangles -- beyond what a row group scans whole
        W, H, nx, ny, F = 1536, 1024, 12, 60, 5
        img = G.lcg_image(W, H, 411)
        sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
        frames = [WL.sin_dst(sp, 5.0 + f, 8 + (f % 4)) for f in range(F)]
        geoms = [WL.piecewise_geom(d) for d in frames]
        ms = WL.src_min(sp)
        offs, total = HG.pack_offsets(geoms)
        d_out = c.alloc(total)
        try:
            c.set_image(img)
            c.set_option("min_row_groups", 0); c.set_option("patch", 0); c.set_option("tile", -1)
            c.piecewise_set_mesh(sp, tris, ms[0], ms[1])
            want = [O.warp_inverse_piecewise(sp, frames[f], tris, img, ms[0], ms[1], *geoms[f]) for f in range(F)]
            for opt in (-1, 0):
                c.set_option("self_spans", opt)
                c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                c.warp_inverse_piecewise_frames_device(d_out)
                c.warp_inverse_piecewise_frames_device(d_out)          # (queued twice: the band counters clean themselves)
                c.sync()
                assert c.last_piecewise_kernel() == 1 and c.last_piecewise_self() == (1 if opt else 0), (opt, c.last_piecewise_kernel(), c.last_piecewise_self())
                for f in range(F):
                    g = geoms[f]
                    assert np.array_equal(c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4), want[f]), ("row bands", opt, f)
            assert c.redone_frames() == 0
        finally:
            c.set_option("patch", -1); c.set_option("self_spans", -1)
            c.free(d_out)
        # (c) 320 x 1 cells, 7 small frames: the host skips its per-triangle walk for such sets and guesses ~63 spans per row, every row is
        #     crossed by ~640 -> the one-row self-span blocks (255) overflow -> frames flagged -> map path; then row lists for this mesh
        c.set_option("patch", 0); c.set_option("compact", 0); c.set_option("self_spans", 1); c.set_option("min_row_groups", 1 << 30)
        W, H, F = 2560, 40, 7
        img = G.lcg_image(W, H, 62)
        sp, tris = WL.grid_points(W, H, 320, 1), WL.grid_triangles(320, 1)
        frames = []
        for f in range(F):
            d = sp.copy(); d[1::2] *= np.float32(1.25 + 0.05 * f)
            frames.append(d)
        geoms = [WL.piecewise_geom(d) for d in frames]
        ms = WL.src_min(sp)
        offs, total = HG.pack_offsets(geoms)
        c.set_image(img)
        c.piecewise_set_mesh(sp, tris, ms[0], ms[1])
        want = [O.warp_inverse_piecewise(sp, frames[f], tris, img, ms[0], ms[1], *geoms[f]) for f in range(F)]
        d_out = c.alloc(total)
        try:
            r0 = c.redone_frames()
            for rnd, self_expected in ((0, 1), (1, 0)):
                c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                c.warp_inverse_piecewise_frames_device(d_out)
                c.sync()
                assert c.last_piecewise_self() == self_expected, (rnd, c.last_piecewise_self())
                for f in range(F):
                    g = geoms[f]
                    assert np.array_equal(c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4), want[f]), ("overflow", rnd, f)
                if rnd == 0:
                    assert c.redone_frames() == r0 + F               # every frame flagged, redone through the materialised map
        finally:
            c.free(d_out)
    finally:
        c.close()


def test_reference_state_entry_points_match_oracle_on_fresh_seeds():
    """hg_warp_inverse_piecewise_state / hg_warp_forward_piecewise_state (the reference's loops over its STALE caches)
    against the oracle's own pieces on random inputs: matrices solved from one pair of point sets, the map rasterised from ANOTHER point set /
    triangle list / window (smaller and larger than the source bounding box the forward loop indexes it with, offset windows, folded meshes),
    an empty map, and a map that names more triangles than there are matrices (-> HG_ERR_RANGE, the reference's TypeError)."""
    rng = np.random.default_rng(20250)
    c = HG.Context(0)
    try:
        n_fwd = n_inv = n_range = 0
        for trial in range(70):
            W, H = int(rng.integers(40, 300)), int(rng.integers(30, 200))
            nx, ny = int(rng.integers(1, 7)), int(rng.integers(1, 6))
            img = G.lcg_image(W, H, 900 + trial)
            sp = WL.grid_points(W, H, nx, ny).reshape(-1, 2).astype(np.float64)
            sp = (sp * rng.uniform(0.6, 1.0) + rng.uniform(0, 12, 2)).astype(np.float32)          # source mesh somewhere inside the image
            tris = WL.grid_triangles(nx, ny)

            def points(scale_lo, scale_hi):
                jit = rng.uniform(0, 0.3)
                d = (sp.astype(np.float64) + rng.uniform(-jit, jit, sp.shape) * [W / nx, H / ny]) * rng.uniform(scale_lo, scale_hi, 2) + rng.uniform(-15, 25, 2)
                if rng.random() < 0.3:
                    d[rng.integers(0, d.shape[0])] += rng.uniform(-40, 40, 2)                       # a fold
                return d.astype(np.float32).ravel()

            dst_pm, dst_map, dst_now = points(0.5, 1.6), points(0.5, 1.8), points(0.7, 1.1)
            mats = HG.solve_affine_triangles(sp.ravel(), dst_pm, tris)
            assert _nan_eq(mats.ravel(), O.piecewise_matrices(sp.ravel(), dst_pm, tris).ravel())
            mm = [int(v) for v in O.minmax_xy(sp.ravel())]
            c.set_image(img)
            # --- forward loop over a stale INVERSE map (what a forward warp after an inverse one reads)
            gm = WL.piecewise_geom(dst_map)                          # the window of the inverse warp that left the map
            mw, mh, myo = (0, 0, 0) if trial % 9 == 0 else (gm[2], gm[3], gm[1])
            tris_map = tris[:3 * max(1, tris.size // 3 - int(rng.integers(0, 3)))] if trial % 4 == 1 else tris
            geom = WL.piecewise_geom(dst_now)
            held = O.build_tri_map(dst_map, tris_map, mw, myo, mw * mh) if mw * mh > 0 else np.zeros(0, np.int16)
            cells = max(mm[2] - mm[0], 0) * max(mm[3] - mm[1], 0)
            eff = np.full(max(cells, 1), -1, np.int16)
            eff[:min(held.size, cells)] = held[:min(held.size, cells)]
            n_mats = tris.size // 3 if trial % 7 else max(1, tris.size // 3 - 2)                   # sometimes fewer matrices than the map's ids
            too_few = bool(eff[:min(held.size, cells)].max(initial=-1) >= n_mats)
            try:
                got = c.warp_forward_piecewise_state(mats[:n_mats], dst_map, tris_map, mw, mh, myo, mm[0], mm[1], mm[2], mm[3], geom)
                assert not too_few, trial
                want = O.warp_forward_piecewise(eff, mats[:n_mats], img, mm[0], mm[1], mm[2], mm[3], *geom)
                assert np.array_equal(got, want), (trial, "forward state")
                n_fwd += 1
            except HG.HgError as e:
                assert too_few and e.code == 6, (trial, str(e))
                n_range += 1
            # --- inverse loop with stale matrices over the map of the current destiny points
            gi = WL.piecewise_geom(dst_now)
            imap = O.build_tri_map(dst_now, tris_map, gi[2], gi[1], gi[2] * gi[3])
            too_few = bool(imap.max(initial=-1) >= n_mats)
            try:
                got = c.warp_inverse_piecewise_state(mats[:n_mats], dst_now, tris_map, mm[0], mm[1], gi)
                assert not too_few, trial
                inv = np.stack([O.inverse_affine(m) for m in mats[:n_mats]])
                want = O.warp_inverse_piecewise_loop(imap, inv, img, mm[0], mm[1], *gi)
                assert np.array_equal(got, want), (trial, "inverse state")
                n_inv += 1
            except HG.HgError as e:
                assert too_few and e.code == 6, (trial, str(e))
                n_range += 1
        assert n_fwd >= 40 and n_inv >= 40 and n_range >= 3, (n_fwd, n_inv, n_range)
        # the usual entry points of the context are untouched by the state calls
        sp0, tr0 = WL.grid_points(128, 96, 4, 3), WL.grid_triangles(4, 3)
        dp0 = WL.sin_dst(sp0, 3.0, 8)
        g0 = WL.piecewise_geom(dp0)
        img0 = G.lcg_image(128, 96, 4)
        c.set_image(img0)
        c.piecewise_set_mesh(sp0, tr0, 0, 0)
        c.piecewise_prepare(dp0, g0)
        first = c.warp_inverse_piecewise()
        c.warp_forward_piecewise_state(HG.solve_affine_triangles(sp0, dp0, tr0), dp0, tr0, g0[2], g0[3], g0[1], 0, 0, 128, 96, g0)
        assert np.array_equal(c.warp_inverse_piecewise(), first) and np.array_equal(first, O.warp_inverse_piecewise(sp0, dp0, tr0, img0, 0, 0, *g0))
    finally:
        c.close()


def test_reference_state_forms_with_more_than_32767_triangles():
    """The state forms over a mesh whose ids do not fit the reference's Int16Array (A-Q9): 134 x 133 cells = 35 644 triangles; ids from 32 768
    on wrap to negative values, which both loops read as "no triangle".  Forward form over the stale inverse map (wider than the source bounding
    box's row length, so rows of the held map straddle the forward index), inverse form over the current map; both against the oracle's pieces."""
    W, H, nx, ny = 420, 400, 134, 133
    img = G.lcg_image(W, H, 91)
    sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
    assert tris.size // 3 > 32767
    dst_map = (WL.sin_dst(sp, 3.0, 8).reshape(-1, 2) * np.float32([1.12, 1.05]) + np.float32([4, 2])).astype(np.float32).ravel()
    dst_now = (WL.sin_dst(sp, 2.0, 9).reshape(-1, 2) * np.float32([0.97, 0.99])).astype(np.float32).ravel()
    mats = HG.solve_affine_triangles(sp, dst_now, tris)
    mm = [int(v) for v in O.minmax_xy(sp)]
    gm, geom = WL.piecewise_geom(dst_map), WL.piecewise_geom(dst_now)
    assert gm[2] > mm[2] - mm[0]                                  # the held map's rows are longer than the forward index's
    held = O.build_tri_map(dst_map, tris, gm[2], gm[1], gm[2] * gm[3])
    assert held.min() < -1                                        # wrapped ids are in the map
    cells = (mm[2] - mm[0]) * (mm[3] - mm[1])
    eff = np.full(cells, -1, np.int16)
    eff[:min(held.size, cells)] = held[:min(held.size, cells)]
    with HG.Context(0) as c:
        c.set_image(img)
        got = c.warp_forward_piecewise_state(mats, dst_map, tris, gm[2], gm[3], gm[1], mm[0], mm[1], mm[2], mm[3], geom)
        assert np.array_equal(got, O.warp_forward_piecewise(eff, mats, img, mm[0], mm[1], mm[2], mm[3], *geom))
        imap = O.build_tri_map(dst_now, tris, geom[2], geom[1], geom[2] * geom[3])
        assert imap.min() < -1
        got = c.warp_inverse_piecewise_state(mats, dst_now, tris, mm[0], mm[1], geom)
        inv = np.stack([O.inverse_affine(m) for m in mats])
        assert np.array_equal(got, O.warp_inverse_piecewise_loop(imap, inv, img, mm[0], mm[1], *geom))
        # the documented limits are string errors, not pixels
        far = dst_map.copy(); far[0] = np.float32(2.0 ** 25)
        with pytest.raises(HG.HgError) as e:
            c.warp_forward_piecewise_state(mats, far, tris, gm[2], gm[3], gm[1], mm[0], mm[1], mm[2], mm[3], geom)
        assert e.value.code == 1                                  # HG_ERR_INVALID
        with pytest.raises(HG.HgError) as e:
            c.warp_forward_piecewise_state(mats, dst_map, tris, gm[2], gm[3], (1 << 26) + 1, mm[0], mm[1], mm[2], mm[3], geom)
        assert e.value.code == 1


def test_forward_state_form_checks_the_held_map_for_a_blank_window_too():
    """A forward frame with a blank window (:440) does not stop the reference's loop :955-969: it walks the source bounding box over the held map
    and throws where a cell names a matrix that does not exist.  hg_warp_forward_piecewise_state therefore checks the ids BEFORE it returns for an
    empty window (HG_ERR_RANGE -> the class's TypeError); with every id in range nothing is written and the call succeeds."""
    W, H, nx, ny = 96, 80, 4, 3
    img = G.lcg_image(W, H, 17)
    sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
    dst = (WL.sin_dst(sp, 2.0, 8).reshape(-1, 2) * np.float32(1.2)).astype(np.float32).ravel()
    gm = WL.piecewise_geom(dst)
    mats = HG.solve_affine_triangles(sp, dst, tris)
    mm = [int(v) for v in O.minmax_xy(sp)]
    T = tris.size // 3
    with HG.Context(0) as c:
        c.set_image(img)
        for geom in ((0, 0, 0, 0), (3, 2, 0, 17), (5, 5, 40, 0)):
            out = c.warp_forward_piecewise_state(mats, dst, tris, gm[2], gm[3], gm[1], mm[0], mm[1], mm[2], mm[3], geom)
            assert out.size == 0
            with pytest.raises(HG.HgError) as e:
                c.warp_forward_piecewise_state(mats[:2], dst, tris, gm[2], gm[3], gm[1], mm[0], mm[1], mm[2], mm[3], geom)     # (the cells the loop reads hold ids beyond 1)
            assert e.value.code == 6                              # HG_ERR_RANGE
        # ... and the context goes on as usual
        geom = WL.piecewise_geom(dst)
        held = O.build_tri_map(dst, tris, gm[2], gm[1], gm[2] * gm[3])
        cells = (mm[2] - mm[0]) * (mm[3] - mm[1])
        eff = np.full(cells, -1, np.int16); eff[:min(held.size, cells)] = held[:min(held.size, cells)]
        got = c.warp_forward_piecewise_state(mats, dst, tris, gm[2], gm[3], gm[1], mm[0], mm[1], mm[2], mm[3], geom)
        assert np.array_equal(got, O.warp_forward_piecewise(eff, mats, img, mm[0], mm[1], mm[2], mm[3], *geom))


def test_general_path_between_two_banded_fast_path_sets():
    """A context that ran the self-span path WITH candidate bands (a mesh beyond 256 triangles on k_pw_patch<SELF>), then a frame set
    the fast kernels do not take (source minimum beyond 2^22: k_pw_fused through k_tri_setup alone) with MORE frames and TALLER windows
    than the band buffers were laid out for, then the banded fast path again: k_tri_setup of the general path must not file triangles
    into the old band buffers (round-4 advisor finding), and every step equals the oracle."""
    c = HG.Context(0)
    try:
        W, H, nx, ny = 768, 512, 20, 16                         # 640 triangles
        img = G.lcg_image(W, H, 77)
        sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
        ms = WL.src_min(sp)
        c.set_image(img)
        c.set_option("min_row_groups", 0); c.set_option("patch", 1); c.set_option("self_spans", 1); c.set_option("tile", 0)    # (tile 0: this test is about k_pw_patch's bands)

        def step(F, scale, msx, msy, expect_self):
            frames = [(WL.sin_dst(sp, 5.0 + f, 8 + (f % 3)).reshape(-1, 2) * np.float32(scale)).astype(np.float32).ravel() for f in range(F)]
            geoms = [WL.piecewise_geom(d) for d in frames]
            offs, total = HG.pack_offsets(geoms)
            d_out = c.alloc(total)
            try:
                c.piecewise_set_mesh(sp, tris, msx, msy)
                c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                c.warp_inverse_piecewise_frames_device(d_out)
                c.sync()
                assert c.last_piecewise_self() == expect_self, (F, scale, msx, c.last_piecewise_kernel())
                for f in range(F):
                    g = geoms[f]
                    want = O.warp_inverse_piecewise(sp, frames[f], tris, img, msx, msy, *g)
                    assert np.array_equal(c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4), want), (F, scale, msx, f)
            finally:
                c.free(d_out)

        step(2, 1.0, ms[0], ms[1], 1)                           # fast path, candidate bands sized for 2 frames of ~520 rows
        assert c.last_piecewise_kernel() == 3
        step(5, 2.5, -(1 << 22) - 8, ms[1], 0)                  # general path: 5 frames of ~1300 rows
        assert c.last_piecewise_kernel() == 4
        step(2, 1.0, ms[0], ms[1], 1)                           # and back
        step(3, 1.7, ms[0], ms[1], 1)
        assert c.redone_frames() == 0
    finally:
        c.close()


def test_flag_word_is_armed_whatever_the_upload_option_was_when_the_run_was_queued():
    """hg_sync skips the status ring when the page-locked flag word is clear: the kernels must set that word for every queued run,
    also for runs queued while option "upload_kernel" was 0 and synced after it went back to 1 (round-4 advisor finding).  A frame
    with a NaN vertex is flagged (irregular) and must come back redone = equal to the oracle; and an irregular frame teaches the
    layout policy nothing: the mesh keeps the self-span path."""
    c = HG.Context(0)
    try:
        W, H, nx, ny, F = 640, 384, 6, 4, 10
        img = G.lcg_image(W, H, 5)
        sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
        ms = WL.src_min(sp)
        frames = [WL.sin_dst(sp, 4.0 + f, 8) for f in range(F)]
        geoms = [WL.piecewise_geom(d) for d in frames]
        frames[3] = frames[3].copy(); frames[3][4] = np.nan       # an irregular frame: a NaN x over finite rows (its window was derived before)
        offs, total = HG.pack_offsets(geoms)
        c.set_image(img); c.piecewise_set_mesh(sp, tris, ms[0], ms[1])
        c.set_option("min_row_groups", 0)
        d_out = c.alloc(total)
        try:
            want = [O.warp_inverse_piecewise(sp, frames[f], tris, img, ms[0], ms[1], *geoms[f]) for f in range(F)]

            def check():
                for f in range(F):
                    g = geoms[f]
                    assert np.array_equal(c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4), want[f]), f

            c.set_option("upload_kernel", 0)
            c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
            c.warp_inverse_piecewise_frames_device(d_out)          # queued, not synced
            c.set_option("upload_kernel", 1)
            c.sync()
            assert c.redone_frames() == 1
            check()
            c.set_option("self_spans", 1)
            for rep in range(2):
                c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                c.warp_inverse_piecewise_frames_device(d_out)
                c.sync()
                assert c.last_piecewise_self() == 1, "an irregular frame must not disable the self-span path"
                check()
            assert c.redone_frames() == 3
        finally:
            c.free(d_out)
    finally:
        c.close()


def test_fresh_point_sets_queue_without_settling_and_redo_from_their_own_set(ctx):
    """The reference's loop uploads new destination points before every warp (test/benchmark.js:107-110).  Here sets and runs
    are queued back to back with no sync in between, into different outputs; set 1 is a mesh too dense for the row lists (its
    frames are flagged and redone through the map path at hg_sync -- by then two newer sets have been uploaded, so the redo has
    to come from the staged copy of ITS set).  Every output matches the oracle; the layout estimate is walked once per shape."""
    W, H, nx, ny = 512, 96, 8, 4
    img = G.lcg_image(W, H, 77)
    sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
    ms = WL.src_min(sp)
    sets = [[WL.sin_dst(sp, 3.0 + s, 8 + f) for f in range(3)] for s in range(4)]
    # set 1: fold the mesh so that every row is crossed by far more spans than the lists hold (-> flagged, redone)
    folded = []
    for f in range(3):
        d = sets[1][f].copy().reshape(-1, 2)
        d[:, 0] = (d[:, 0] * 37.0) % 97.0 + 0.25 * f          # x scrambled into [0, 97): hundreds of sliver spans per row
        folded.append(d.ravel().astype(np.float32))
    sets[1] = folded
    ctx.set_image(img)
    ctx.piecewise_set_mesh(sp, tris, ms[0], ms[1])
    geoms = [[WL.piecewise_geom(d) for d in fs] for fs in sets]
    packs = [HG.pack_offsets(g) for g in geoms]
    outs = [ctx.alloc(total) for _, total in packs]
    try:
        ctx.sync()
        walks0, redone0 = ctx.layout_walks(), ctx.redone_frames()
        for s in range(4):
            ctx.piecewise_set_frames(np.concatenate(sets[s]), geoms[s], packs[s][0])
            ctx.warp_inverse_piecewise_frames_device(outs[s])
        ctx.sync()
        for s in range(4):
            for f, g in enumerate(geoms[s]):
                want = O.warp_inverse_piecewise(sp, sets[s][f], tris, img, ms[0], ms[1], *g)
                got = ctx.to_host(outs[s], g[2] * g[3] * 4, packs[s][0][f]).reshape(g[3], g[2], 4)
                assert np.array_equal(got, want), (s, f)
        assert ctx.layout_walks() - walks0 <= 4
        # the same sets again, resident shapes: sets 0, 2, 3 share a window shape -> at most one more walk each for 0 and 1
        w1 = ctx.layout_walks()
        for _ in range(3):
            for s in (0, 2, 3):
                ctx.piecewise_set_frames(np.concatenate(sets[s]), geoms[s], packs[s][0])
                ctx.warp_inverse_piecewise_frames_device(outs[s])
        ctx.sync()
        assert ctx.layout_walks() - w1 <= 1, ctx.layout_walks() - w1
        for s in (0, 2, 3):
            for f, g in enumerate(geoms[s]):
                want = O.warp_inverse_piecewise(sp, sets[s][f], tris, img, ms[0], ms[1], *g)
                got = ctx.to_host(outs[s], g[2] * g[3] * 4, packs[s][0][f]).reshape(g[3], g[2], 4)
                assert np.array_equal(got, want), (s, f)
        # many steps alternating between windows of different heights: the two counter sets stay clean (nothing is flagged, nothing redone)
        tall = [WL.sin_dst(sp, 9.0, 8 + f) for f in range(3)]
        gt = [WL.piecewise_geom(d) for d in tall]
        pt = HG.pack_offsets(gt)
        assert any(a[3] != b[3] for a, b in zip(gt, geoms[0]))
        big = ctx.alloc(max(pt[1], packs[0][1]))
        try:
            ctx.sync()
            r1 = ctx.redone_frames()
            for it in range(120):
                which = (it % 3) != 0
                ctx.piecewise_set_frames(np.concatenate(tall if which else sets[0]), gt if which else geoms[0], pt[0] if which else packs[0][0])
                ctx.warp_inverse_piecewise_frames_device(big)
            ctx.sync()
            assert ctx.redone_frames() == r1, "frames were flagged: a counter set was left dirty"
            for f, g in enumerate(gt):                         # the last step (it = 119) warped the tall set
                want = O.warp_inverse_piecewise(sp, tall[f], tris, img, ms[0], ms[1], *g)
                assert np.array_equal(ctx.to_host(big, g[2] * g[3] * 4, pt[0][f]).reshape(g[3], g[2], 4), want), ("alternating", f)
        finally:
            ctx.free(big)
    finally:
        for o in outs:
            ctx.free(o)


def test_redo_of_an_earlier_step_never_lands_on_a_later_steps_frames():
    """One output buffer reused step after step (bench.py --points fresh): step 1 warps a mesh whose rows carry 1100 spans (always
    beyond the row lists: flagged, to be redone through the map path at hg_sync), step 2 -- queued before any sync -- warps a
    degenerate point set (every x equal: no spans, a blank frame) into the SAME buffer.  The deferred redo of step 1 must not
    put its frame over step 2's."""
    n, W2, H2 = 1100, 2400, 8
    img2 = G.lcg_image(W2, H2, 10)
    xs = np.linspace(0, W2, n + 1)
    sp2 = np.stack([np.repeat(xs, 2), np.tile([0.0, H2], n + 1)], 1).astype(np.float32).ravel()
    tr2 = np.array([[2 * i, 2 * i + 2, 2 * i + 1] for i in range(n)], np.uint32).ravel()
    dp_a = sp2.copy(); dp_a[1::2] *= 1.5
    dp_b = dp_a.copy(); dp_b[0::2] = 5.0
    mm, md = O.minmax_xy(sp2), O.minmax_xy(dp_a)
    g = (int(md[0]), int(md[1]), int(md[2] - md[0]), int(md[3] - md[1]))
    want_a = O.warp_inverse_piecewise(sp2, dp_a, tr2, img2, int(mm[0]), int(mm[1]), *g)
    want_b = O.warp_inverse_piecewise(sp2, dp_b, tr2, img2, int(mm[0]), int(mm[1]), *g)
    assert want_a.any() and not want_b.any()
    c = HG.Context(0)
    try:
        c.set_image(img2)
        c.piecewise_set_mesh(sp2, tr2, int(mm[0]), int(mm[1]))
        d_out = c.alloc(g[2] * g[3] * 4)
        c.piecewise_set_frames(dp_a, [g], [0]); c.warp_inverse_piecewise_frames_device(d_out)
        c.piecewise_set_frames(dp_b, [g], [0]); c.warp_inverse_piecewise_frames_device(d_out)
        c.sync()
        assert c.redone_frames() >= 1                         # step 1 WAS flagged ...
        assert np.array_equal(c.to_host(d_out, g[2] * g[3] * 4).reshape(g[3], g[2], 4), want_b)     # ... and step 2's frame stands
        # the other order: the flagged step is the last one into the buffer, so its redo is what the buffer must hold
        c.piecewise_set_frames(dp_b, [g], [0]); c.warp_inverse_piecewise_frames_device(d_out)
        c.piecewise_set_frames(dp_a, [g], [0]); c.warp_inverse_piecewise_frames_device(d_out)
        c.sync()
        assert np.array_equal(c.to_host(d_out, g[2] * g[3] * 4).reshape(g[3], g[2], 4), want_a)
        c.free(d_out)
    finally:
        c.close()


def test_empty_and_degenerate_inputs(ctx):
    img = G.lcg_image(32, 24, 3)
    ctx.set_image(img)
    # zero-area output window: nothing to do, no crash
    assert ctx.warp_inverse_geometric(0, np.array([1, 0, 0, 1, 0, 0], np.float64), (0, 0, 0, 10)).size == 0
    # mesh with zero triangles: every pixel stays 0
    sp = np.array([0, 0, 31, 0, 0, 23], np.float32)
    ctx.piecewise_set_mesh(sp, np.zeros(0, np.uint32), 0, 0)
    ctx.piecewise_prepare(sp, (0, 0, 31, 23))
    assert not ctx.warp_inverse_piecewise().any()
    # NaN matrix: bounds test fails everywhere
    out = ctx.warp_inverse_geometric(1, np.full(8, np.nan), (0, 0, 16, 8))
    assert out.shape == (8, 16, 4) and not out.any()
    # call-order errors are loud
    c2 = HG.Context(0)
    with pytest.raises(HG.HgError):
        c2.warp_inverse_geometric(0, np.zeros(6), (0, 0, 4, 4))
    c2.close()


# ------------------------------------------------------------------ BASELINE.json full-size configs: size-independent properties

def _full_case(name):
    return next((c for c in GOLD["cases"] if c["name"] == name), None)


@pytest.mark.parametrize("name", ["C2_projective_1080p", "C3_piecewise_4k", "C3_piecewise_4k_5000tri", "C5_piecewise_8k"])
def test_full_size_configs_sha_and_properties(ctx, name):
    case = _full_case(name)
    if case is None:
        pytest.skip("full-size goldens not generated")
    w = case["warps"][0]
    out = hip_run_warp(ctx, case, 0)
    assert G.sha256(out) == w["out"]["sha"]                       # bit-exact against the reference at full size
    img = G.case_images(case)[G.warp_image_key(case, 0)]
    # property 1: hit count on an all-255 source == the reference's N_hit
    ctx.set_image(np.full_like(img, 255))
    if w["transform"] == "piecewiseaffine":
        solid = ctx.warp_inverse_piecewise()
    else:
        sp, dp = G.f32_from_bits(w["srcPoints"]), G.f32_from_bits(w["dstPoints"])
        solid = ctx.warp_inverse_geometric(1, HG.solve_projective(dp, sp), (w["xOff"], w["yOff"], w["objW"], w["objH"]))
    assert int(np.count_nonzero(solid[..., 3] == 255)) == w["nhit"]
    assert set(np.unique(solid)) <= {0, 255}
    # property 2: every output pixel is either 0 or an exact copy of some source pixel (gather-only, no blending):
    # the multiset of non-zero output pixels is a sub-multiset-by-value of the source pixels
    src_vals = np.unique(img.reshape(-1, 4).view(np.uint32))
    out_vals = np.unique(out.reshape(-1, 4).view(np.uint32))
    assert np.isin(out_vals[out_vals != 0], src_vals).all()
    # property 3: linearity in the source under XOR with a constant image on hit pixels (pure copy)
    flip = img ^ 0x5A
    ctx.set_image(flip)
    if w["transform"] == "piecewiseaffine":
        out2 = ctx.warp_inverse_piecewise()
    else:
        out2 = ctx.warp_inverse_geometric(1, HG.solve_projective(dp, sp), (w["xOff"], w["yOff"], w["objW"], w["objH"]))
    hit = solid[..., 3] == 255
    assert np.array_equal(out2[hit], out[hit] ^ 0x5A) and not out2[~hit].any()


@pytest.mark.parametrize("W,H,F", [(480, 270, 16), (3840, 2160, 3)])
def test_c4_face_mesh_batch_matches_oracle(ctx, W, H, F):
    """C4: 68 landmarks triangulated by the host Delaunay (hg_triangulate), landmarks orbiting their rest
    position; a small analogue over a quarter of the orbit and three frames at the full 4K size, against the oracle."""
    img = G.lcg_image(W, H, 68)
    sp = WL.face_mesh(W, H)
    tris = HG.triangulate(sp)
    assert 100 <= tris.size // 3 <= 130
    seq = WL.face_frames(sp, W, 64 if W < 1000 else 512)
    frames = [seq[(f * 5) % len(seq)] for f in range(F)]
    geoms = [WL.piecewise_geom(d) for d in frames]
    msx, msy = WL.src_min(sp)
    ctx.set_image(img)
    ctx.piecewise_set_mesh(sp, tris, msx, msy)
    offs, total = HG.pack_offsets(geoms)
    d_out = ctx.alloc(total)
    try:
        ctx.piecewise_set_frames(np.concatenate(frames), geoms, offs)
        ctx.warp_inverse_piecewise_frames_device(d_out)
        ctx.sync()
        for f in range(F):
            g = geoms[f]
            got = ctx.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
            want = O.warp_inverse_piecewise(sp, frames[f], tris, img, msx, msy, *g)
            assert np.array_equal(got, want), f
            assert (got[..., 3] > 0).mean() > 0.5          # the mesh really covers most of its bounding box
    finally:
        ctx.free(d_out)


def test_queued_runs_settle_in_call_order():
    """Asynchronous `_frames_device` calls are queued back to back (no host round trip); frames a fused run flags (here: row
    lists overflow on the first calls of a fresh context) are redone at hg_sync into the output of the call that flagged
    them.  70 calls > the status ring (64) into three rotating output buffers."""
    c = HG.Context(0)
    try:
        n, W2, H2 = 300, 1200, 12
        img = G.lcg_image(W2, H2, 21)
        xs = np.linspace(0, W2, n + 1)
        sp = np.stack([np.repeat(xs, 2), np.tile([0.0, H2], n + 1)], 1).astype(np.float32).ravel()
        tr = np.array([[2 * i, 2 * i + 2, 2 * i + 1] for i in range(n)] + [[2 * i + 1, 2 * i + 2, 2 * i + 3] for i in range(n)], np.uint32).ravel()
        frames = [sp.copy(), sp.copy()]
        frames[0][1::2] *= 1.5
        frames[1][1::2] *= 2.0
        geoms = [WL.piecewise_geom(d) for d in frames]
        ms = WL.src_min(sp)
        want = [O.warp_inverse_piecewise(sp, frames[f], tr, img, ms[0], ms[1], *geoms[f]) for f in range(2)]
        c.set_image(img)
        c.piecewise_set_mesh(sp, tr, ms[0], ms[1])
        offs, total = HG.pack_offsets(geoms)
        bufs = [c.alloc(total) for _ in range(3)]
        try:
            c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
            for k in range(70):
                c.warp_inverse_piecewise_frames_device(bufs[k % 3])
            c.sync()
            for b in bufs:
                for f in range(2):
                    g = geoms[f]
                    got = c.to_host(b, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
                    assert np.array_equal(got, want[f])
        finally:
            for b in bufs:
                c.free(b)
    finally:
        c.close()


def test_dense_sheared_mesh_takes_the_patch_kernel():
    """A mesh with ~150 spans per row and shear ~1 (C5's regime at a size the oracle does in a blink): the host picks the 2-D gather
    kernels -- since round 6 k_pw_tile (8-row tiles, source-row-following runs) where the mesh is steeply sheared or packs more than two
    spans into a 64-pixel block, k_pw_patch (4-row groups, one matrix record per triangle of the group) for a flat dense mesh; a dense
    mesh too narrow for its bins and a sparse mesh stay on k_pw_rows.  All bit-exact against the oracle, three frames with different windows per batch."""
    c = HG.Context(0)
    c.set_option("min_row_groups", 0)            # (these frame sets are small: without this they would all run one row per workgroup)
    try:
        # (dense sheared, more spans per row than a tile holds -> k_pw_patch; dense sheared within the tile's limits -> k_pw_tile (5; round 6, EXPERIMENTS.md R6.4:
        #  before that k_pw_patch); dense flat -> k_pw_patch since round 4 (its self-span form needs no producer kernel);
        #  dense rows too narrow for its bins -> k_pw_rows one row per workgroup; sparse -> 4-row groups; very dense, ~300 spans per
        #  row -> k_pw_patch in its global-record variant)
        for (W, H, nx, ny, A, want_kernel) in [(1600, 150, 56, 3, 14.0, 3), (3200, 150, 32, 3, 40.0, 5), (3200, 150, 17, 3, 0.5, 3), (640, 150, 60, 3, 0.5, 2),
                                               (640, 150, 8, 3, 18.0, 1), (3000, 100, 110, 2, 5.0, 3)]:
            img = G.lcg_image(W, H, 31)
            sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
            frames = [WL.sin_dst(sp, A, 8 + f) for f in range(3)]
            frames[1] = (frames[1].reshape(-1, 2) * np.float32([1.1, 0.9]) + np.float32([7, 3])).ravel()
            geoms = [WL.piecewise_geom(d) for d in frames]
            ms = WL.src_min(sp)
            c.set_image(img)
            c.piecewise_set_mesh(sp, tris, ms[0], ms[1])
            offs, total = HG.pack_offsets(geoms)
            d_out = c.alloc(total)
            try:
                c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                for _ in range(3):                           # queued runs: counters are cleaned by the kernel itself
                    c.warp_inverse_piecewise_frames_device(d_out)
                c.sync()
                assert c.last_piecewise_kernel() == want_kernel, (nx, A, c.last_piecewise_kernel())
                assert c.redone_frames() == 0                 # nothing went through the map-path fallback
                for f, g in enumerate(geoms):
                    got = c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
                    assert np.array_equal(got, O.warp_inverse_piecewise(sp, frames[f], tris, img, ms[0], ms[1], *g)), (nx, A, f)
            finally:
                c.free(d_out)
    finally:
        c.close()


def test_patch_kernel_limits_fall_back_and_stay_off():
    """(a) 56 triangle columns squeezed into 300 pixels: ~150 spans per row but ~30 per 64-pixel bin, beyond the 8 slots of
    k_pw_patch's bins: handled inside the kernel (those blocks test the row's whole list), no fallback.  (b) 110 columns:
    ~220 spans per row, beyond its LDS budget: the kernel only flags the frames, hg_sync redoes them through the
    materialised map, and the context then stays on k_pw_rows.  The host's estimates would keep both meshes on k_pw_rows,
    so the patch kernel is forced."""
    for nx, expect_redone in ((56, False), (110, True)):
        c = HG.Context(0)
        c.set_option("patch", 1)
        c.set_option("min_row_groups", 0)
        try:
            W, H, ny, A = 300, 120, 3, 6.0
            img = G.lcg_image(W, H, 41)
            sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
            frames = [WL.sin_dst(sp, A, 8 + f) for f in range(2)]
            geoms = [WL.piecewise_geom(d) for d in frames]
            ms = WL.src_min(sp)
            want = [O.warp_inverse_piecewise(sp, frames[f], tris, img, ms[0], ms[1], *geoms[f]) for f in range(2)]
            c.set_image(img)
            c.piecewise_set_mesh(sp, tris, ms[0], ms[1])
            offs, total = HG.pack_offsets(geoms)
            d_out = c.alloc(total)
            try:
                kernels = []
                for _ in range(2):
                    c.piecewise_set_frames(np.concatenate(frames), geoms, offs)
                    c.warp_inverse_piecewise_frames_device(d_out)
                    c.sync()
                    kernels.append(c.last_piecewise_kernel())
                    for f, g in enumerate(geoms):
                        got = c.to_host(d_out, g[2] * g[3] * 4, offs[f]).reshape(g[3], g[2], 4)
                        assert np.array_equal(got, want[f]), (nx, f)
                if expect_redone:
                    assert kernels[0] == 3 and kernels[1] in (1, 2) and c.redone_frames() == 2, (kernels, c.redone_frames())
                else:
                    assert kernels == [3, 3] and c.redone_frames() == 0, (kernels, c.redone_frames())
            finally:
                c.free(d_out)
        finally:
            c.close()


def test_shared_reciprocal_division_is_ieee_division():
    """The projective kernel divides both numerators by one refined reciprocal when the host has shown the frame's operands
    stay in the plain range; that sequence must give the bits of IEEE division there: 2^29 random / edge-mantissa triples."""
    c = HG.Context(0)
    try:
        assert c.selftest_division(1 << 29, seed=20260928) == 0
        assert c.selftest_division(1 << 26, seed=7) == 0
    finally:
        c.close()


def test_huge_source_row_is_rejected_not_wrapped(ctx):
    """A sliver triangle (0.0002 px wide in the output, 2e6 px tall in the source) whose left-overshoot pixels map to source
    rows around 300 * 2^24: the reference rejects them (srcY >= height, :1047); an implementation that leaves the upper y
    bound to a 32-bit byte offset would wrap them back into the image.  Also a projective frame with the same property."""
    W = H = 64
    img = G.lcg_image(W, H, 5)
    sp = np.array([20, 5, 20, 60, 20.003, 2015599.75, 0, 0, 63, 0, 0, 63], np.float32)
    dp = np.array([10.5001, 0, 10.5001, 63, 10.4999, 0, 0, 0, 63, 0, 0, 63], np.float32)
    tris = np.array([3, 4, 5, 0, 1, 2], np.uint32)
    ms, md = O.minmax_xy(sp), O.minmax_xy(dp)
    geom = (int(md[0]), int(md[1]), int(md[2] - md[0]), int(md[3] - md[1]))
    want, wmap, _, winv = O.warp_inverse_piecewise(sp, dp, tris, img, int(ms[0]), int(ms[1]), *geom, taps=True)
    sliver = wmap.reshape(geom[3], geom[2]) == 1
    iv = winv[1].astype(np.float64)
    ys, xs = np.nonzero(sliver)
    sy = iv[1] * (xs + geom[0]) + iv[3] * (ys + geom[1]) + iv[5]
    assert sliver.sum() > 20 and sy.min() > 2.0 ** 32 and np.all(want[sliver] == 0)      # the case is what it claims to be
    ctx.set_image(img)
    ctx.piecewise_set_mesh(sp, tris, int(ms[0]), int(ms[1]))
    ctx.piecewise_prepare(dp, geom)
    assert np.array_equal(ctx.warp_inverse_piecewise(), want)
    assert np.array_equal(ctx.warp_inverse_piecewise_via_map(), want)
    # geometric kernels: sy = x * 2^24 * 300 / 10 ... an affine / projective matrix that sends column 10 to row 300 * 2^24
    for kind, m in ((0, [1, 503316480.0, 0, 1, 0, 0]), (1, [1, 0, 0, 503316480.0, 1, 0, 0, 0])):
        m = np.array(m, np.float64)
        g = (0, 0, 32, 8)
        wantg = O.warp_inverse_geometric(kind, m, img, *g)
        assert np.array_equal(ctx.warp_inverse_geometric(kind, m, g), wantg), kind


def test_far_away_and_absurd_vertices(ctx):
    """A destiny vertex a million pixels away makes a triangle two million rows tall: the rasterisers only walk the rows that
    can reach the window, the result is still the oracle's.  Infinite or > 2^24 coordinates are refused (the reference's
    row loop would not terminate); NaN stays legal (draws nothing)."""
    import time
    W, H = 96, 64
    img = G.lcg_image(W, H, 12)
    sp, tris = WL.grid_points(W, H, 3, 2), WL.grid_triangles(3, 2)
    dp = WL.sin_dst(sp, 3.0, 8).reshape(-1, 2).copy()
    dp[5] = [40.0, 1.0e6]
    dp[7] = [-3.0e5, -2.0e6]
    dp = dp.astype(np.float32).ravel()
    geom = (0, 0, 96, 70)                                    # (the window is the caller's: here the sane part of the mesh)
    ms = WL.src_min(sp)
    want, wmap, _, _ = O.warp_inverse_piecewise(sp, dp, tris, img, ms[0], ms[1], *geom, taps=True)
    ctx.set_image(img)
    ctx.piecewise_set_mesh(sp, tris, ms[0], ms[1])
    ctx.piecewise_prepare(dp, geom)
    t0 = time.perf_counter()
    assert np.array_equal(ctx.warp_inverse_piecewise(), want)
    assert np.array_equal(ctx.get_tri_map(), wmap)
    assert np.array_equal(ctx.warp_inverse_piecewise_via_map(), want)
    assert time.perf_counter() - t0 < 5.0
    for bad in (np.inf, -np.inf, 3.0e7, -1.0e30):
        d2 = dp.copy(); d2[3] = bad
        with pytest.raises(HG.HgError):
            ctx.piecewise_prepare(d2, geom)
        s2 = sp.copy(); s2[2] = bad
        with pytest.raises(HG.HgError):
            ctx.piecewise_set_mesh(s2, tris, ms[0], ms[1])
    d3 = dp.copy(); d3[3] = np.nan
    ctx.piecewise_set_mesh(sp, tris, ms[0], ms[1])
    ctx.piecewise_prepare(d3, geom)
    assert np.array_equal(ctx.warp_inverse_piecewise(), O.warp_inverse_piecewise(sp, d3, tris, img, ms[0], ms[1], *geom))


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_multi_device_batch_with_flagged_frames(devices):
    """hg_multi_* queues every device's D2H copies before it waits for any device; frames the fused path only flagged are redone by
    that device's settlement afterwards and copied AGAIN.  A mesh whose rows always overflow the span lists (1100 thin triangles):
    every frame takes that road, to host arrays and as resident frames, twice in a row (capacity policies change after a redo)."""
    n, W2, H2, F = 1100, 2400, 8, 5
    img = G.lcg_image(W2, H2, 10)
    xs = np.linspace(0, W2, n + 1)
    sp = np.stack([np.repeat(xs, 2), np.tile([0.0, H2], n + 1)], 1).astype(np.float32).ravel()
    tris = np.array([[2 * i, 2 * i + 2, 2 * i + 1] for i in range(n)], np.uint32).ravel()
    frames = []
    for f in range(F):
        d = sp.copy(); d[1::2] *= 1.2 + 0.2 * f; d[0::2] += 3.0 * f
        frames.append(d)
    geoms = [WL.piecewise_geom(d) for d in frames]
    mm = O.minmax_xy(sp)
    want = [O.warp_inverse_piecewise(sp, frames[f], tris, img, int(mm[0]), int(mm[1]), *geoms[f]) for f in range(F)]
    assert all(w.any() for w in want)
    with HG.Multi(devices) as m:
        m.set_image(img)
        m.piecewise_set_mesh(sp, tris, int(mm[0]), int(mm[1]))
        for rep in range(2):
            outs = [np.zeros_like(w) for w in want]
            m.warp_piecewise_batch(np.concatenate(frames), geoms, [o.ctypes.data for o in outs])
            for f in range(F):
                assert np.array_equal(outs[f], want[f]), ("host", rep, f)
            m.warp_piecewise_batch(np.concatenate(frames), geoms)
            for f in range(F):
                assert np.array_equal(m.frame_to_host(f), want[f]), ("resident", rep, f)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_multi_device_batch(devices):
    """hg_multi_*: the batch caller loop over a device list from one host thread.  [0] is the degenerate single-GPU case;
    [0, 0(, 0)] puts several contexts on the one GPU of this box, which runs the REAL multi-device code path -- partition,
    scatter + all-gather fan-out of the source through hipMemcpyPeerAsync, per-device launches, per-device D2H -- with the
    device talking to itself.  Every frame against the oracle; host outputs (pinned and pageable) and resident frames."""
    W, H, nx, ny, F = 352, 208, 8, 5, 7
    img = G.lcg_image(W, H, 321)
    sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
    frames = [WL.sin_dst(sp, 4.0 + f, 8 + (f % 4)) for f in range(F)]
    geoms = [WL.piecewise_geom(d) for d in frames]
    mm = O.minmax_xy(sp)
    want = [O.warp_inverse_piecewise(sp, frames[f], tris, img, int(mm[0]), int(mm[1]), *geoms[f]) for f in range(F)]
    with HG.Multi(devices) as m:
        assert m.device_count() == len(devices)
        m.set_image(img)
        m.piecewise_set_mesh(sp, tris, int(mm[0]), int(mm[1]))
        m.warp_piecewise_batch(np.concatenate(frames), geoms)                      # frames stay resident
        seen = set()
        for f in range(F):
            dev, ptr, n = m.frame(f)
            assert n == want[f].nbytes and (dev, ptr) not in seen
            seen.add((dev, ptr))
            assert dev == [k for k in range(len(devices)) if HG.multi_partition(F, len(devices), k)[0] <= f][-1]
            assert np.array_equal(m.frame_to_host(f), want[f]), ("resident", f)
        pinned = [HG.PinnedBuffer(w.nbytes) for w in want]
        try:
            m.warp_piecewise_batch(np.concatenate(frames), geoms, [p.ptr for p in pinned])
            for f in range(F):
                assert np.array_equal(pinned[f].array.reshape(want[f].shape), want[f]), ("pinned", f)
        finally:
            for p in pinned:
                p.free()
        outs = [np.zeros_like(w) for w in want]
        m.warp_piecewise_batch(np.concatenate(frames), geoms, [o.ctypes.data for o in outs])
        for f in range(F):
            assert np.array_equal(outs[f], want[f]), ("pageable", f)
        # projective frames over the same device list, solved per device
        s4 = WL.corners(W, H)
        d4s = [WL.projective_dst(W, H, 0.03 * k) for k in range(5)]
        pg = [tuple(int(v) for v in O.transform_limits(1, O.projective_from_squares(s4, d4), W, H)) for d4 in d4s]
        m.warp_geometric_batch(1, np.concatenate(d4s), np.tile(s4, 5), pg)
        for k in range(5):
            assert np.array_equal(m.frame_to_host(k), O.warp_inverse_geometric(1, HG.solve_projective(d4s[k], s4), img, *pg[k])), ("projective", k)
        # fewer frames than devices (some devices idle) and an empty window among them
        few = [frames[0], frames[1]]
        fgeoms = [geoms[0], (geoms[1][0], geoms[1][1], 0, geoms[1][3])]
        m.warp_piecewise_batch(np.concatenate(few), fgeoms)
        assert np.array_equal(m.frame_to_host(0), want[0]) and m.frame(1)[2] == 0
        # a second image of another size through the same object (buffers regrow, aliases are re-attached)
        img2 = G.lcg_image(W + 64, H + 32, 322)
        m.set_image(img2)
        m.warp_piecewise_batch(np.concatenate(frames), geoms)
        for f in (0, F - 1):
            assert np.array_equal(m.frame_to_host(f), O.warp_inverse_piecewise(sp, frames[f], tris, img2, int(mm[0]), int(mm[1]), *geoms[f])), ("regrown", f)
        # hg_multi_frame refuses frames outside the last batch, and a failed batch leaves no stale partition behind
        with pytest.raises(HG.HgError):
            m.frame(F)
        with pytest.raises(HG.HgError):
            m.frame(-1)


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_multi_device_batch_with_one_source_per_frame(devices):
    """SURVEY.md §8e's "replicas only" branch through hg_multi_*: every frame has its own source image (the video loop,
    README.md:121-137); each device uploads only the images of its block, nothing is exchanged.  Piecewise and projective
    frames, resident and to host, every frame against the oracle; then back to a shared source."""
    W, H, nx, ny, F = 288, 176, 6, 4, 7
    imgs = [G.lcg_image(W, H, 500 + f) for f in range(F)]
    sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
    frames = [WL.sin_dst(sp, 3.0 + f, 8 + (f % 4)) for f in range(F)]
    geoms = [WL.piecewise_geom(d) for d in frames]
    mm = O.minmax_xy(sp)
    want = [O.warp_inverse_piecewise(sp, frames[f], tris, imgs[f], int(mm[0]), int(mm[1]), *geoms[f]) for f in range(F)]
    with HG.Multi(devices) as m:
        m.piecewise_set_mesh(sp, tris, int(mm[0]), int(mm[1]))
        m.warp_piecewise_batch_images(np.concatenate(frames), geoms, imgs)         # no hg_multi_set_image needed
        for f in range(F):
            assert np.array_equal(m.frame_to_host(f), want[f]), ("resident", f)
        outs = [np.zeros_like(w) for w in want]
        m.warp_piecewise_batch_images(np.concatenate(frames), geoms, imgs, [o.ctypes.data for o in outs])
        for f in range(F):
            assert np.array_equal(outs[f], want[f]), ("host", f)
        s4 = WL.corners(W, H)
        d4s = [WL.projective_dst(W, H, 0.03 * k) for k in range(5)]
        pg = [tuple(int(v) for v in O.transform_limits(1, O.projective_from_squares(s4, d4), W, H)) for d4 in d4s]
        m.warp_geometric_batch_images(1, np.concatenate(d4s), np.tile(s4, 5), pg, imgs[:5])
        for k in range(5):
            assert np.array_equal(m.frame_to_host(k), O.warp_inverse_geometric(1, HG.solve_projective(d4s[k], s4), imgs[k], *pg[k])), ("projective", k)
        # the shared-source batch refuses to run on the leftovers of the per-frame sources ...
        with pytest.raises(HG.HgError):
            m.warp_piecewise_batch(np.concatenate(frames), geoms)
        # ... and works again after hg_multi_set_image
        m.set_image(imgs[2])
        m.warp_piecewise_batch(np.concatenate(frames), geoms)
        assert np.array_equal(m.frame_to_host(2), want[2])
