"""Loader for tests/golden/golden.json + golden_blobs.bin (written by tests/golden/gen_golden.mjs)."""
import functools
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "golden")


@functools.lru_cache(maxsize=1)
def load():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        g = json.load(f)
    with open(os.path.join(GOLDEN_DIR, "golden_blobs.bin"), "rb") as f:
        g["_blobs"] = f.read()
    return g


def blob(ref, dtype):
    b = load()["_blobs"][ref["off"]: ref["off"] + ref["len"]]
    return np.frombuffer(b, dtype=dtype).copy()


def f32_from_bits(x):
    """list of uint32 bit patterns, or a {off,len,dtype:'f32bits'} blob ref -> float32 array"""
    if isinstance(x, dict):
        return blob(x, np.uint32).view(np.float32)
    return np.array(x, dtype=np.uint32).view(np.float32)


def f64_from_hex(xs):
    return np.array([int(h, 16) for h in xs], dtype=np.uint64).view(np.float64)


def bits32(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def bits64(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def lcg_image(w, h, seed):
    """(cached: the 8K image takes seconds to build and the full-size cases all share seed 1; callers get a private copy only
    for small images, the big ones are returned read-only)"""
    a = _lcg_image_cached(int(w), int(h), int(seed))
    return a if a.nbytes > (1 << 22) else a.copy()


@functools.lru_cache(maxsize=3)
def _lcg_image_cached(w, h, seed):
    a = _lcg_image(w, h, seed)
    a.setflags(write=False)
    return a


def _lcg_image(w, h, seed):
    """SURVEY.md §8d synthetic RGBA: s = s*1664525 + 1013904223 mod 2^32; byte = s >> 24 (vectorised by LCG jump-ahead)."""
    n = w * h * 4
    # s_k = a^k * s0 + c * (a^k - 1)/(a - 1)  (mod 2^32): build a^k and the geometric sum by doubling blocks
    a, c = np.uint64(1664525), np.uint64(1013904223)
    mask = np.uint64(0xFFFFFFFF)
    mul = np.empty(n, dtype=np.uint64)
    add = np.empty(n, dtype=np.uint64)
    mul[0], add[0] = a, c
    filled = 1
    while filled < n:
        m = min(filled, n - filled)
        # state after (filled + j + 1) steps = apply (filled) steps then (j+1) steps
        am, cm = mul[filled - 1], add[filled - 1]
        mul[filled:filled + m] = (mul[:m] * am) & mask
        add[filled:filled + m] = (mul[:m] * cm + add[:m]) & mask
        filled += m
    s = (mul * np.uint64(seed) + add) & mask
    return (s >> np.uint64(24)).astype(np.uint8).reshape(h, w, 4)


def case_images(case):
    out = {}
    for k, v in (case.get("images") or {}).items():
        if v.get("solid"):
            out[k] = np.full((v["h"], v["w"], 4), 255, np.uint8)
        else:
            out[k] = lcg_image(v["w"], v["h"], v["seed"])
    return out


def case_triangles(case):
    t = case.get("triangles")
    if t is None:
        return np.zeros(0, np.uint32)
    if isinstance(t, dict):
        return blob(t["u32blob"], np.uint32)
    return np.array(t, dtype=np.uint32)


def warp_image_key(case, warp_index):
    """Which image the k-th warp record of the case operated on (same rule as the generator; a warpBatch op stands for one warp() per
    destiny set, a throwing warp() still occupies a record)."""
    last = None
    k = -1
    ow = case.get("opWarps")
    for i, op in enumerate(case["script"]):
        name, a = op[0], op[1:]
        key = None
        if name == "setSourcePoints" and len(a) > 1:
            key = a[1]
        elif name == "setImage":
            key = a[0]
        elif name == "setReferencePoints" and len(a) > 2:
            key = a[2]
        elif name == "warp" and len(a) > 0:
            key = a[0]
        if key is not None:
            last = key
        if name in ("warp", "warpBatch"):
            first, n = ow[str(i)] if ow and str(i) in ow else (k + 1, 1 if name == "warp" else len(a[0]))
            if first <= warp_index < first + n:
                return last
            k = first + n - 1
    raise IndexError(warp_index)


def is_pixel_warp(w):
    """A warp record with pixels to compare: not one whose warp() threw in the reference, nor a blank 1 x 1 frame (:440: no output
    window yet, e.g. right after setSourcePoints reset it).  Those are checked by the JS replay only."""
    return "throws" not in w and isinstance(w.get("objW"), (int, float)) and isinstance(w.get("objH"), (int, float)) and w["objW"] * w["objH"] >= 1


def _ints(x, dtype):
    return blob(x, dtype) if isinstance(x, dict) else np.array(x, dtype=dtype)


def warp_triangles(case, w):
    """The triangle list in force at that warp (setTriangles may have replaced the case's own)."""
    return _ints(w["trisNow"], np.uint32) if "trisNow" in w else case_triangles(case)


def stale_map_def(w):
    """(points f32, triangles u32, width, height, y_off) of the map the shared field held when a forward loop over stale state ran."""
    d = w["stale"]["mapDef"]
    # (an inverse map built before any output window existed has width = height = null: `new Int16Array(null * null)` is empty)
    return f32_from_bits(d["pts"]), _ints(d["tris"], np.uint32), int(d["width"] or 0), int(d["height"] or 0), int(d["yOff"] or 0)


def cases(filter_fn=None):
    return [c for c in load()["cases"] if filter_fn is None or filter_fn(c)]
