"""The C ABI's threading contract (include/hgwarp.h:16): "a ctx is not thread-safe, DISTINCT ctxs are independent".
Two hg_ctx on device 0, each driven by its own host thread (ctypes releases the GIL around every call, so the two threads really are
inside libhgwarp.so at the same time), each replaying a different set of the reference's golden warps for a few hundred calls: every
result must still be the reference's bytes.  Plus the error slots: a failure on one ctx / thread must not show up in the other's
hg_last_error.  Run with `pytest -m gpu` on an MI355X."""
import os
import threading

import numpy as np
import pytest

from hgtest import golden as G
from hgtest import hip
from hgtest import workloads as WL

pytestmark = pytest.mark.gpu

HG = hip.load()
GOLD = G.load()


def _small_warps(max_pixels=1 << 19):
    out = []
    for c in GOLD["cases"]:
        for k, w in enumerate(c["warps"]):
            if G.is_pixel_warp(w) and "sha" in w.get("out", {}) and w["out"]["w"] * w["out"]["h"] <= max_pixels:
                out.append((c, k))
    return out


def _run_set(ctx, work, iters, errors, tag, barrier):
    from test_gpu_parity import hip_run_warp           # the parity suite's own dispatcher (same calls as every golden test)
    try:
        barrier.wait(timeout=60)
        n = 0
        while n < iters:
            for case, k in work:
                w = case["warps"][k]
                out = hip_run_warp(ctx, case, k)
                if G.sha256(out) != w["out"]["sha"]:
                    errors.append(f"{tag}: {case['name']}#{k} differs from the reference after {n} calls")
                    return
                n += 1
                if n >= iters:
                    break
    except Exception as e:                               # noqa: BLE001 -- reported through the list, the thread must not die silently
        errors.append(f"{tag}: {type(e).__name__}: {e}")


def test_two_contexts_on_two_host_threads_replay_different_goldens():
    warps = _small_warps()
    assert len(warps) >= 40
    sets = [warps[0::2], warps[1::2]]                    # disjoint: the threads never run the same case at the same time
    paths = [{c["warps"][k]["path"] for c, k in s} for s in sets]
    for p in paths:                                      # both threads see inverse + forward loops, geometric + piecewise
        assert {"_inversePiecewiseAffineWarp", "_inverseGeometricWarp", "_piecewiseAffineWarp"} <= p, p
    ctxs = [HG.Context(0), HG.Context(0)]
    try:
        errors, barrier = [], threading.Barrier(2)
        ts = [threading.Thread(target=_run_set, args=(ctxs[i], sets[i], int(os.environ.get("HG_THREAD_ITERS", "400")), errors, f"thread {i}", barrier)) for i in range(2)]
        for t in ts: t.start()
        for t in ts: t.join(timeout=900)
        assert not any(t.is_alive() for t in ts), "a thread is stuck inside the library"
        assert not errors, errors
    finally:
        for c in ctxs: c.close()


def test_queued_batches_from_two_threads_do_not_mix():
    """The queued form (frame sets staged in each context's own ring, outputs left on the device): thread 0 queues C4-like batches,
    thread 1 a 10 x 10 grid, 60 steps each without a settlement in between; every frame equals the single-threaded result."""
    W, H = 640, 360
    img = WL.lcg_image(W, H, 7)

    def workload(nx, ny, F, seed):
        sp, tris = WL.grid_points(W, H, nx, ny), WL.grid_triangles(nx, ny)
        dps = [WL.sin_dst(sp, 3.0 + f, 5 + seed) for f in range(F)]
        geoms = [WL.piecewise_geom(dp) for dp in dps]
        return sp, tris, np.stack(dps), geoms

    def run(ctx, wl, steps, barrier=None):
        sp, tris, dps, geoms = wl
        msx, msy = WL.src_min(sp)
        ctx.set_image(img)
        ctx.piecewise_set_mesh(sp, tris, msx, msy)
        offs, total = HG.pack_offsets(geoms)
        d_out = ctx.alloc(total)
        if barrier: barrier.wait(timeout=60)
        for _ in range(steps):
            ctx.piecewise_set_frames(dps, geoms, offs)
            ctx.warp_inverse_piecewise_frames_device(d_out)
        ctx.sync()
        got = ctx.to_host(d_out, total).copy()
        ctx.free(d_out)
        return got

    wls = [workload(6, 5, 8, 0), workload(10, 10, 5, 1)]
    with HG.Context(0) as c:
        want = [run(c, wl, 1) for wl in wls]
    ctxs = [HG.Context(0), HG.Context(0)]
    try:
        res, errors, barrier = [None, None], [], threading.Barrier(2)

        def body(i):
            try: res[i] = run(ctxs[i], wls[i], 60, barrier)
            except Exception as e: errors.append(f"thread {i}: {e}")    # noqa: BLE001

        ts = [threading.Thread(target=body, args=(i,)) for i in range(2)]
        for t in ts: t.start()
        for t in ts: t.join(timeout=600)
        assert not errors, errors
        for i in range(2):
            assert res[i] is not None and np.array_equal(res[i], want[i]), f"thread {i}: batch differs from the single-threaded run"
            assert ctxs[i].redone_frames() == 0
    finally:
        for c in ctxs: c.close()


def test_last_error_is_per_context_and_per_thread():
    L = HG.lib()
    a, b = HG.Context(0), HG.Context(0)
    try:
        with pytest.raises(HG.HgError):
            a.set_option("xcc", 3)                       # not a power of two: HG_ERR_INVALID on ctx a
        msg_a = L.hg_last_error(a._h)
        assert msg_a and b"xcc" in msg_a
        assert not L.hg_last_error(b._h)                 # ctx b has seen no failure
        seen = {}

        def other():                                     # another thread: its context-less slot is its own
            seen["before"] = L.hg_last_error(None)
            import ctypes as C
            h = C.c_void_p()
            seen["code"] = L.hg_create(4096, C.byref(h))         # no such device
            seen["after"] = L.hg_last_error(None)

        main_before = L.hg_last_error(None)
        t = threading.Thread(target=other); t.start(); t.join(timeout=60)
        assert seen["code"] != 0 and seen["after"]
        assert not seen["before"]                        # a fresh thread starts with an empty slot, whatever this thread's holds
        assert L.hg_last_error(None) == main_before      # ... and its failure did not touch this thread's
        assert L.hg_last_error(a._h) == msg_a and not L.hg_last_error(b._h)
        # ctx b still works after a's failure
        b.set_image(WL.lcg_image(64, 48, 1))
    finally:
        a.close(); b.close()
