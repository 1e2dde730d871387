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
