#!/usr/bin/env python3
"""Host Delaunay (hg_triangulate) on hostile inputs: duplicates, collinear sets, lattices (all cocircular), tiny / huge / mixed
magnitudes, NaN / Inf (must be refused, not crash).  Checks structure (ids in range, no degenerate triangle, Euler count on
general-position sets) and, run under tools/run_asan.sh, memory safety.   python tools/fuzz_delaunay.py [trials]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hgtest import hip      # noqa: E402

HG = hip.load()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(2024)
refused = 0
for t in range(trials):
    n = int(rng.integers(1, 400))
    kind = t % 8
    if kind == 0:
        p = rng.uniform(0, 1000, (n, 2))
    elif kind == 1:
        p = rng.integers(0, 6, (n, 2)).astype(float) * 17          # lattice with many duplicates
    elif kind == 2:
        x = rng.uniform(0, 100, n); p = np.stack([x, 2 * x + 1], 1)  # collinear
    elif kind == 3:
        p = np.repeat(rng.uniform(0, 10, (1, 2)), n, 0)              # all the same point
    elif kind == 4:
        p = rng.uniform(0, 1, (n, 2)) * 10.0 ** rng.integers(-30, 30, (n, 2))   # wild magnitudes
    elif kind == 5:
        a = rng.uniform(0, 2 * np.pi, n); p = np.stack([np.cos(a), np.sin(a)], 1) * 500 + 500   # cocircular
    elif kind == 6:
        p = rng.uniform(0, 1000, (n, 2)); p[rng.integers(0, n)] = [np.nan, 1.0]
    else:
        p = rng.uniform(0, 1000, (n, 2)); p[rng.integers(0, n)] = [np.inf, -np.inf]
    p32 = p.astype(np.float32).ravel()
    try:
        tri = HG.triangulate(p32)
    except HG.HgError:
        refused += 1
        assert kind in (4, 6, 7), f"trial {t} kind {kind}: refused a finite input"
        continue
    assert kind not in (6, 7), f"trial {t}: non-finite input accepted"
    assert tri.size % 3 == 0 and (tri.size == 0 or int(tri.max()) < n)
    T = tri.reshape(-1, 3)
    assert np.all((T[:, 0] != T[:, 1]) & (T[:, 1] != T[:, 2]) & (T[:, 0] != T[:, 2])), f"trial {t}: triangle with a repeated vertex"
    assert T.shape[0] <= 2 * n
print(f"delaunay fuzz ok: {trials} trials, {refused} refused (non-finite / overflowing inputs)")
