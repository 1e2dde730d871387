"""Pins the CPU oracle (oracle/hg_oracle.c) against the golden vectors produced by running the reference's own
Homography.js (tests/golden/gen_golden.mjs) and against the reference's known-answer PNG pair.  CPU only."""
import os

import numpy as np
import pytest

from hgtest import golden as G
from hgtest import oracle as O

GOLD = G.load()


# ------------------------------------------------------------------ per-function vectors

def test_lcg_image_matches_c_and_numpy():
    for (w, h, seed) in [(7, 5, 1), (64, 33, 12345), (400, 400, 7)]:
        assert np.array_equal(G.lcg_image(w, h, seed), O.lcg_image(w, h, seed))


def test_js_round_vectors():
    for v in GOLD["func"]["round"]:
        x = G.f64_from_hex([v["x"]])[0]
        want = G.f64_from_hex([v["r"]])[0]
        got = O.js_round(x)
        if np.isnan(want):
            assert np.isnan(got)
        else:
            assert got == want, (x, got, want)       # sign of zero is not observable downstream


def test_affine_from_triangles_bit_exact():
    for v in GOLD["func"]["affine"]:
        got = O.affine_from_triangles(G.f32_from_bits(v["src"]), G.f32_from_bits(v["dst"]))
        want = G.f32_from_bits(v["out"])
        assert _same_f32(got, want), (v, got, want)


def test_inverse_affine_bit_exact():
    for v in GOLD["func"]["inv_affine"]:
        got = O.inverse_affine(G.f32_from_bits(v["m"]))
        assert _same_f32(got, G.f32_from_bits(v["out"]))


def test_projective_from_squares_bit_exact():
    for v in GOLD["func"]["projective"]:
        got = O.projective_from_squares(G.f32_from_bits(v["src"]), G.f32_from_bits(v["dst"]))
        want = G.f64_from_hex(v["out"])
        assert _same_f64(got, want), (got, want)


def test_fill_triangle_maps_bit_exact():
    for v in GOLD["func"]["fill"]:
        m = np.full(v["width"] * v["rows"], -1, np.int16)
        O.fill_triangle(G.f32_from_bits(v["tri"]), v["idx"], v["width"], v["yoff"], m)
        assert np.array_equal(m, np.array(v["map"], np.int16)), v


def test_transform_limits_and_minmax():
    for v in GOLD["func"]["limits"]:
        if v["kind"] == "affine":
            m = G.f32_from_bits(v["m"]).astype(np.float64)
            got = O.transform_limits(0, m, v["w"], v["h"])
        else:
            got = O.transform_limits(1, G.f64_from_hex(v["m"]), v["w"], v["h"])
        assert _same_f64(got, G.f64_from_hex(v["out"]))
    for v in GOLD["func"]["minmax"]:
        assert _same_f64(O.minmax_xy(G.f32_from_bits(v["p"])), G.f64_from_hex(v["out"]))


def _same_f32(a, b):
    a, b = G.bits32(a), G.bits32(b)
    nan = np.isnan(a.view(np.float32)) & np.isnan(b.view(np.float32))
    zero = (a.view(np.float32) == 0) & (b.view(np.float32) == 0)
    return bool(np.all((a == b) | nan | zero))


def _same_f64(a, b):
    a, b = G.bits64(a), G.bits64(b)
    nan = np.isnan(a.view(np.float64)) & np.isnan(b.view(np.float64))
    zero = (a.view(np.float64) == 0) & (b.view(np.float64) == 0)
    return bool(np.all((a == b) | nan | zero))


# ------------------------------------------------------------------ end-to-end warps (resolved state -> oracle -> golden output)

SLOW = {"C2_projective_1080p", "C3_piecewise_4k", "C3_piecewise_4k_5000tri", "C5_piecewise_8k", "C3_batch_4k", "C5_batch_8k", "C4_orbit_4k"}


def _warp_params():
    ps = []
    for c in GOLD["cases"]:
        for k, w in enumerate(c["warps"]):
            if not G.is_pixel_warp(w):              # the reference's warp() threw there, or returned its blank 1 x 1 frame (JS replay)
                continue
            marks = [pytest.mark.slow] if c["name"] in SLOW else []
            ps.append(pytest.param(c["name"], k, id=f"{c['name']}#{k}", marks=marks))
    return ps


def oracle_run_warp(case, k):
    """Feeds the golden's resolved low-level state of warp #k to the oracle; returns (rgba, map|None, fwd|None, inv|None)."""
    w = case["warps"][k]
    img = G.case_images(case)[G.warp_image_key(case, k)]
    assert img.shape[1] == w["W"] and img.shape[0] == w["H"]
    path = w["path"]
    if w["transform"] == "piecewiseaffine" and w.get("stale"):
        return oracle_run_stale_warp(case, w, img)
    if w["transform"] == "piecewiseaffine":
        sp, dp = G.f32_from_bits(w["srcPoints"]), G.f32_from_bits(w["dstPoints"])
        tris = G.warp_triangles(case, w)
        if path == "_inversePiecewiseAffineWarp":
            return O.warp_inverse_piecewise(sp, dp, tris, img, w["minSrcX"], w["minSrcY"], w["xOff"], w["yOff"], w["objW"], w["objH"], taps=True)
        fwd = O.piecewise_matrices(sp, dp, tris)
        mw, mh = w["maxSrcX"] - w["minSrcX"], w["maxSrcY"] - w["minSrcY"]
        fmap = O.build_tri_map(sp, tris, mw, w["minSrcY"], mw * mh)
        out = O.warp_forward_piecewise(fmap, fwd, img, w["minSrcX"], w["minSrcY"], w["maxSrcX"], w["maxSrcY"], w["xOff"], w["yOff"], w["objW"], w["objH"])
        return out, fmap, fwd, None
    kind = 0 if w["transform"] == "affine" else 1
    if path == "_inverseGeometricWarp":
        # Q8: the inverse matrix is re-solved from the swapped point sets, :994
        sp, dp = G.f32_from_bits(w["srcPoints"]), G.f32_from_bits(w["dstPoints"])
        m = O.affine_from_triangles(dp, sp).astype(np.float64) if kind == 0 else O.projective_from_squares(dp, sp)
        want_m = G.f32_from_bits(w["invMatrix"]["f32"]).astype(np.float64) if kind == 0 else G.f64_from_hex(w["invMatrix"]["f64"])
        assert _same_f64(m, want_m)
        return O.warp_inverse_geometric(kind, m, img, w["xOff"], w["yOff"], w["objW"], w["objH"]), None, None, None
    m = G.f32_from_bits(w["matrix"]["f32"]).astype(np.float64) if kind == 0 else G.f64_from_hex(w["matrix"]["f64"])
    return O.warp_forward_geometric(kind, m, img, w["xOff"], w["yOff"], w["objW"], w["objH"]), None, None, None


def stale_effective_map(w):
    """What _piecewiseAffineWarp :957 reads when the shared field holds something else than the forward map of the current mesh
    (SURVEY.md Appendix A-Q12): the stale map rasterised from ITS point set with ITS geometry, indexed (maxSrcX - minSrcX) cells per
    row; cells past its end are `undefined` (-1 here).  Returns (the map as the reference held it, the cells the loop can index)."""
    pts, mtris, mw, mh, myoff = G.stale_map_def(w)
    own = mw * mh if mw > 0 and mh > 0 else 0
    held = O.build_tri_map(pts, mtris, mw, myoff, own)
    cells = max(w["maxSrcX"] - w["minSrcX"], 0) * max(w["maxSrcY"] - w["minSrcY"], 0)
    eff = np.full(max(cells, 1), -1, np.int16)
    eff[:min(own, cells)] = held[:min(own, cells)]
    return held, eff


def oracle_run_stale_warp(case, w, img):
    """A warp whose loop read stale caches: the matrices as the reference held them (`fwd` blob) and, for the forward loop, the map the
    shared field held -- composed from the oracle's own pieces (rasteriser, inverseAffineMatrix, the two loops)."""
    fwd = G.blob(w["fwd"], np.float32).reshape(-1, 6)
    assert fwd.shape[0] == w["stale"]["nMats"]
    if w["path"] == "_inversePiecewiseAffineWarp":
        dp, tris = G.f32_from_bits(w["dstPoints"]), G.warp_triangles(case, w)
        n = w["objW"] * w["objH"]
        map_ = O.build_tri_map(dp, tris, w["objW"], w["yOff"], n)
        assert map_.max(initial=-1) < fwd.shape[0]
        inv = np.stack([O.inverse_affine(m) for m in fwd]) if len(fwd) else np.zeros((0, 6), np.float32)
        out = O.warp_inverse_piecewise_loop(map_, inv, img, w["minSrcX"], w["minSrcY"], w["xOff"], w["yOff"], w["objW"], w["objH"])
        return out, map_, fwd, inv
    held, eff = stale_effective_map(w)
    out = O.warp_forward_piecewise(eff, fwd, img, w["minSrcX"], w["minSrcY"], w["maxSrcX"], w["maxSrcY"], w["xOff"], w["yOff"], w["objW"], w["objH"])
    return out, held, fwd, None


@pytest.mark.parametrize("name,k", _warp_params())
def test_oracle_reproduces_reference_warp(name, k):
    case = next(c for c in GOLD["cases"] if c["name"] == name)
    w = case["warps"][k]
    out, map_, fwd, inv = oracle_run_warp(case, k)
    assert out.shape[1] == w["out"]["w"] and out.shape[0] == w["out"]["h"]
    if "blob" in w["out"]:
        want = G.blob(w["out"]["blob"], np.uint8).reshape(out.shape)
        assert np.array_equal(out, want), f"{np.count_nonzero(np.any(out != want, axis=2))} pixels differ"
    assert G.sha256(out) == w["out"]["sha"]
    if map_ is not None:
        assert map_.size == w["map"]["len"]
        if "blob" in w["map"]:
            assert np.array_equal(map_, G.blob(w["map"]["blob"], np.int16))
        assert G.sha256(map_) == w["map"]["sha"]
    for got, key, shakey in ((fwd, "fwd", "fwdSha"), (inv, "inv", "invSha")):
        if got is None:
            continue
        if key in w:
            assert _same_f32(got.ravel(), G.blob(w[key], np.float32))
        if not np.isnan(got).any():                 # NaN payload/sign bits are not observable in JS
            assert G.sha256(got) == w[shakey]


# ------------------------------------------------------------------ the reference's own known-answer fixture (test/nodeTest.js)

def test_known_answer_png_pair():
    Image = pytest.importorskip("PIL.Image")
    d = os.path.join(G.GOLDEN_DIR, "ref_fixture")
    src = np.array(Image.open(os.path.join(d, "testImgLogoBlack.png")).convert("RGBA"))
    want = np.array(Image.open(os.path.join(d, "transformedImage.png")).convert("RGBA"))
    W, H = 400, 400
    # test/nodeTest.js:5-6 (normalised) -> denormalised by _setSrcWidthHeight :652-662 once the image is set
    sp = (np.array([[0, 0], [0, 1], [1, 0], [1, 1]], np.float32) * np.array([W, H], np.float32)).astype(np.float32)
    dp = (np.array([[1 / 10, 1 / 2], [0, 1], [9 / 10, 1 / 2], [1, 1]], np.float32) * np.array([W, H], np.float32)).astype(np.float32)
    fwd = O.projective_from_squares(sp.ravel(), dp.ravel())
    xo, yo, ow, oh = [int(v) for v in O.transform_limits(1, fwd, W, H)]
    assert (xo, yo, ow, oh) == (0, 200, 400, 200)
    inv = O.projective_from_squares(dp.ravel(), sp.ravel())
    out = O.warp_inverse_geometric(1, inv, src, xo, yo, ow, oh)
    assert out.shape == want.shape
    assert np.array_equal(out, want), f"{np.count_nonzero(np.any(out != want, axis=2))} of {ow * oh} pixels differ"


def test_js_oracle_pinned_to_goldens():
    """oracle/hg_oracle_js.mjs (used only to time the algorithm under Node next to the GPU numbers) reproduces the goldens."""
    import json
    import shutil
    import subprocess
    node = shutil.which("node")
    if node is None:
        pytest.skip("node missing")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([node, os.path.join(root, "oracle", "hg_oracle_js.mjs"), "check"], capture_output=True, text=True, timeout=600)
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["failures"] == [] and res["checked"] >= 30 and p.returncode == 0


def test_seam_sensitivity_to_triangle_order_on_c4():
    """What "triangulation parity unpinned" can cost: the triangle SET of a Delaunay triangulation is
    unique for points in general position, the ORDER of the list is an implementation detail of delaunator -- and the reference's
    map is last-writer-wins (:852-858), so order decides who owns every cell that two triangles both cover.  C4 frame 0
    (68-landmark face mesh, 4K) through the oracle with the list as hg_triangulate returns it, reversed, and shuffled:
      * the orbit as SURVEY.md §8d defines it moves every landmark on its own circle of radius 0.02 W (77 px, neighbours are ~270 px
        apart): destination triangles overlap where the mesh folds, and there the order decides between two genuinely different
        source regions -- a few percent of the window change owner and value;
      * the same orbit at a tenth of the radius (no folds): NOTHING changes -- the half-open spans [round(x0), round(x1)) of
        neighbouring triangles tile a row without overlap, so on an unfolded mesh the order of the list is immaterial and only
        the triangle SET (the diagonal chosen for co-circular quads) matters.
    Coverage (which cells have an owner at all) never depends on the order."""
    from hgtest import workloads as WL
    from hgtest import hip
    HG = hip.load()
    W, H = 3840, 2160
    sp = WL.face_mesh(W, H, 68)
    tris = HG.triangulate(sp).reshape(-1, 3)
    ms = WL.src_min(sp)
    img = G.lcg_image(W, H, 1)
    dp0 = WL.face_frames(sp, W, 512)[0]
    gentle = (sp.astype(np.float64) + 0.1 * (dp0.astype(np.float64) - sp.astype(np.float64))).astype(np.float32)
    report = {}
    for label, dp in (("orbit radius 0.02 W", dp0), ("orbit radius 0.002 W", gentle)):
        geom = WL.piecewise_geom(dp)
        base, bmap, _, _ = O.warp_inverse_piecewise(sp, dp, tris.ravel(), img, ms[0], ms[1], *geom, taps=True)
        n = geom[2] * geom[3]
        rng = np.random.default_rng(4)
        worst_map = worst_px = 0
        for order in (np.arange(len(tris))[::-1], rng.permutation(len(tris)), rng.permutation(len(tris))):
            out, omap, _, _ = O.warp_inverse_piecewise(sp, dp, np.ascontiguousarray(tris[order]).ravel(), img, ms[0], ms[1], *geom, taps=True)
            assert np.array_equal(omap >= 0, bmap >= 0)               # same coverage
            inv = np.asarray(order)                                    # id in the permuted list -> id in the original list
            same_owner = np.where(omap >= 0, inv[np.clip(omap, 0, None)], -1) == bmap
            d_map = int(np.count_nonzero(~same_owner))
            d_px = int(np.count_nonzero(np.any(out != base, axis=2)))
            worst_map, worst_px = max(worst_map, d_map), max(worst_px, d_px)
            assert d_px <= d_map                                       # pixels can only differ where the owner differs
        report[label] = (geom, worst_map / n, worst_px / n)
        print(f"C4 frame 0, {len(tris)} triangles, {label} (xOff {geom[0]}, {geom[2]}x{geom[3]}): up to {worst_map} of {n} map cells "
              f"({100.0 * worst_map / n:.3f} %) change owner with the triangle order, {worst_px} pixels ({100.0 * worst_px / n:.3f} %) change value")
    assert 0 < report["orbit radius 0.02 W"][1] <= 0.10 and report["orbit radius 0.02 W"][2] <= 0.10
    assert report["orbit radius 0.002 W"][1] == 0 and report["orbit radius 0.002 W"][2] == 0
