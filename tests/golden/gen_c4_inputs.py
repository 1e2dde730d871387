#!/usr/bin/env python3
"""Inputs of the C4 golden case (68-landmark face mesh, 512-frame orbit) as plain data for gen_golden.mjs.

The mesh comes from homography.js_amd/workloads.py (numpy RNG: not reproducible inside Node) and the triangles from the
library's own host Delaunay (hg_triangulate: CPU only), so they are written once to tests/golden/c4_inputs.json; the
generator injects them into the reference exactly like every other golden case injects its triangles.
    python tests/golden/gen_c4_inputs.py        (needs homography.js_amd/lib/libhgwarp.so; no GPU)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hgtest import hip, workloads as WL      # noqa: E402

FRAMES = [0, 1, 2, 3, 128, 256, 384, 511]

cfg = WL.CONFIGS["C4"]
W, H = cfg["W"], cfg["H"]
sp = WL.face_mesh(W, H, cfg["landmarks"])
tris = hip.load().triangulate(sp)
seq = WL.face_frames(sp, W, cfg["total_frames"])
out = {"W": W, "H": H, "frames": FRAMES, "src": [float(v) for v in sp], "triangles": [int(v) for v in tris],
       "dst": [[float(v) for v in seq[f]] for f in FRAMES],
       "note": "float32 values written as exact doubles; triangles = hg_triangulate(src) (own Delaunay, injected into the reference)"}
with open(os.path.join(ROOT, "tests", "golden", "c4_inputs.json"), "w") as f:
    json.dump(out, f)
print(f"{len(sp) // 2} landmarks, {len(tris) // 3} triangles, {len(FRAMES)} frames")
