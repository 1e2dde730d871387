"""ctypes binding of lib/libhgwarp.so (C ABI: include/hgwarp.h).

Plumbing only: used by tests/, bench.py and __graft_entry__.py.  The product's host side is the drop-in JavaScript
class js/Homography.mjs over the N-API addon; this module exposes the same C entry points to Python 1:1 and adds
nothing on top (no CPU fallback: if the library or a gfx950 GPU is missing, calls raise).

The directory name contains a dot, so load this file by path:
    import importlib.util, sys
    spec = importlib.util.spec_from_file_location("hgwarp", ".../homography.js_amd/hgwarp.py")
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# HGWARP_LIB: another build of the SAME library (the sanitizer build lib/libhgwarp_asan.so, tools/run_asan.sh)
LIB_PATH = os.environ.get("HGWARP_LIB") or os.path.join(HERE, "lib", "libhgwarp.so")

HG_AFFINE, HG_PROJECTIVE = 0, 1

# every symbol include/hgwarp.h declares (tests check that the built library exports all of them)
EXPORTS = [
    "hg_version", "hg_device_count", "hg_create", "hg_create_on_stream", "hg_destroy", "hg_last_error", "hg_sync",
    "hg_device_alloc", "hg_device_free", "hg_copy_to_host", "hg_copy_to_device", "hg_copy_to_host_async", "hg_host_alloc", "hg_host_free", "hg_ctx_device",
    "hg_multi_create", "hg_multi_destroy", "hg_multi_last_error", "hg_multi_device_count", "hg_multi_ctx", "hg_multi_partition", "hg_multi_plan_fanout", "hg_multi_peer_note", "hg_multi_peer_access", "hg_multi_set_image",
    "hg_multi_piecewise_set_mesh", "hg_multi_warp_piecewise_batch", "hg_multi_warp_geometric_batch", "hg_multi_warp_piecewise_batch_images",
    "hg_multi_warp_geometric_batch_images", "hg_multi_frame", "hg_enqueue_copy_to_host", "hg_stream_wait_event",
    "hg_solve_affine", "hg_invert_affine", "hg_solve_projective", "hg_transform_limits", "hg_minmax_xy", "hg_js_round",
    "hg_triangulate",
    "hg_set_image", "hg_set_image_device", "hg_set_images_device",
    "hg_warp_inverse_geometric", "hg_warp_inverse_geometric_device", "hg_geometric_set_frames", "hg_geometric_set_frames_points", "hg_get_geometric_matrices",
    "hg_warp_inverse_geometric_frames_device", "hg_warp_inverse_geometric_batch_device", "hg_pack_offsets",
    "hg_piecewise_set_mesh", "hg_piecewise_prepare", "hg_warp_inverse_piecewise", "hg_warp_inverse_piecewise_device",
    "hg_piecewise_set_frames", "hg_warp_inverse_piecewise_frames_device", "hg_warp_inverse_piecewise_batch_device",
    "hg_get_tri_map", "hg_get_tri_map_fused", "hg_get_matrices", "hg_warp_inverse_piecewise_via_map",
    "hg_warp_forward_geometric", "hg_warp_forward_piecewise", "hg_warp_forward_geometric_device", "hg_warp_forward_geometric_batch_device",
    "hg_warp_forward_piecewise_device", "hg_warp_forward_piecewise_batch_device",
    "hg_solve_affine_triangles", "hg_warp_inverse_piecewise_state", "hg_warp_forward_piecewise_state",
    "hg_upload_on_copy_stream", "hg_fence_copies", "hg_download_behind_warps", "hg_fence_downloads",
    "hg_set_timing", "hg_last_kernel_ms", "hg_kernel_ms_stats", "hg_last_piecewise_kernel", "hg_last_piecewise_variant", "hg_last_piecewise_self", "hg_last_piecewise_flag", "hg_last_forward_kernel", "hg_forward_tiles_admissible", "hg_redone_frames", "hg_layout_walks", "hg_set_option", "hg_xcc_count", "hg_selftest_division", "hg_projective_plain_range", "hg_affine_one_fma_form",
]


class Geom(C.Structure):
    _fields_ = [("x_off", C.c_int32), ("y_off", C.c_int32), ("obj_w", C.c_int32), ("obj_h", C.c_int32)]


class TriMapDef(C.Structure):
    """hg_tri_map_def: what the reference's shared map field was rasterised from (include/hgwarp.h, reference-state forms)."""
    _fields_ = [("points", C.POINTER(C.c_float)), ("n_points", C.c_int), ("triangles", C.POINTER(C.c_uint32)), ("n_triangles", C.c_int),
                ("width", C.c_int32), ("height", C.c_int32), ("y_off", C.c_int32)]


class HgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"hgwarp error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Loads libhgwarp.so (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HgError(-1, f"{LIB_PATH} is missing: build it with `make -C homography.js_amd` (or __graft_entry__.build())")
    L = C.CDLL(LIB_PATH)
    vp, i, sz, d = C.c_void_p, C.c_int, C.c_size_t, C.c_double
    f32p, f64p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    sig = {
        "hg_version": (i, []), "hg_device_count": (i, [C.POINTER(i)]),
        "hg_create": (i, [i, C.POINTER(vp)]), "hg_create_on_stream": (i, [i, vp, C.POINTER(vp)]), "hg_destroy": (None, [vp]),
        "hg_last_error": (C.c_char_p, [vp]), "hg_sync": (i, [vp]),
        "hg_device_alloc": (i, [vp, sz, C.POINTER(vp)]), "hg_device_free": (i, [vp, vp]), "hg_copy_to_host": (i, [vp, vp, vp, sz]), "hg_copy_to_device": (i, [vp, vp, vp, sz]),
        "hg_copy_to_host_async": (i, [vp, vp, vp, sz]), "hg_host_alloc": (i, [sz, C.POINTER(vp)]), "hg_host_free": (i, [vp]), "hg_ctx_device": (i, [vp]),
        "hg_multi_create": (i, [C.POINTER(i), i, C.POINTER(vp)]), "hg_multi_destroy": (None, [vp]), "hg_multi_last_error": (C.c_char_p, [vp]),
        "hg_multi_device_count": (i, [vp]), "hg_multi_ctx": (vp, [vp, i]),
        "hg_multi_partition": (i, [i, i, i, C.POINTER(i), C.POINTER(i)]),
        "hg_multi_plan_fanout": (i, [i, vp, vp, i]), "hg_multi_peer_note": (C.c_char_p, [vp]), "hg_multi_peer_access": (i, [vp, i, i]),
        "hg_multi_set_image": (i, [vp, u8p, i, i]), "hg_multi_piecewise_set_mesh": (i, [vp, f32p, i, C.POINTER(C.c_uint32), i, i, i]),
        "hg_multi_warp_piecewise_batch": (i, [vp, f32p, C.POINTER(Geom), i, C.POINTER(vp)]),
        "hg_multi_warp_geometric_batch": (i, [vp, i, f32p, f32p, C.POINTER(Geom), i, C.POINTER(vp)]),
        "hg_multi_frame": (i, [vp, i, C.POINTER(i), C.POINTER(vp), C.POINTER(sz)]),
        "hg_multi_warp_piecewise_batch_images": (i, [vp, f32p, C.POINTER(Geom), i, C.POINTER(vp), i, i, C.POINTER(vp)]),
        "hg_multi_warp_geometric_batch_images": (i, [vp, i, f32p, f32p, C.POINTER(Geom), i, C.POINTER(vp), i, i, C.POINTER(vp)]),
        "hg_enqueue_copy_to_host": (i, [vp, vp, vp, sz]), "hg_stream_wait_event": (i, [vp, vp]),
        "hg_solve_affine": (i, [f32p, f32p, f32p]), "hg_invert_affine": (i, [f32p, f32p]), "hg_solve_projective": (i, [f32p, f32p, f64p]),
        "hg_transform_limits": (i, [i, f64p, d, d, f64p]), "hg_minmax_xy": (i, [f32p, i, f64p]), "hg_js_round": (d, [d]),
        "hg_triangulate": (i, [f32p, i, C.POINTER(C.c_uint32), i, C.POINTER(i)]),
        "hg_set_image": (i, [vp, u8p, i, i]), "hg_set_image_device": (i, [vp, vp, i, i]),
        "hg_set_images_device": (i, [vp, vp, i, i, i, sz]),
        "hg_warp_inverse_geometric": (i, [vp, i, f64p, Geom, u8p]), "hg_warp_inverse_geometric_device": (i, [vp, i, f64p, Geom, vp]),
        "hg_geometric_set_frames": (i, [vp, i, f64p, C.POINTER(Geom), C.POINTER(sz), i]),
        "hg_geometric_set_frames_points": (i, [vp, i, f32p, f32p, C.POINTER(Geom), C.POINTER(sz), i]),
        "hg_get_geometric_matrices": (i, [vp, f64p, i]),
        "hg_warp_inverse_geometric_frames_device": (i, [vp, vp]),
        "hg_warp_inverse_geometric_batch_device": (i, [vp, i, f64p, C.POINTER(Geom), C.POINTER(sz), i, vp]),
        "hg_pack_offsets": (i, [C.POINTER(Geom), i, C.POINTER(sz), C.POINTER(sz)]),
        "hg_piecewise_set_mesh": (i, [vp, f32p, i, C.POINTER(C.c_uint32), i, i, i]),
        "hg_piecewise_prepare": (i, [vp, f32p, Geom]),
        "hg_warp_inverse_piecewise": (i, [vp, u8p]), "hg_warp_inverse_piecewise_device": (i, [vp, vp]),
        "hg_piecewise_set_frames": (i, [vp, f32p, C.POINTER(Geom), C.POINTER(sz), i]),
        "hg_warp_inverse_piecewise_frames_device": (i, [vp, vp]),
        "hg_warp_inverse_piecewise_batch_device": (i, [vp, f32p, C.POINTER(Geom), C.POINTER(sz), i, vp]),
        "hg_get_tri_map": (i, [vp, C.POINTER(C.c_int16), sz]), "hg_get_tri_map_fused": (i, [vp, C.POINTER(C.c_int16), sz]),
        "hg_get_matrices": (i, [vp, f32p, f32p]), "hg_warp_inverse_piecewise_via_map": (i, [vp, u8p]),
        "hg_last_piecewise_kernel": (i, [vp]), "hg_last_piecewise_variant": (i, [vp]), "hg_last_piecewise_self": (i, [vp]), "hg_last_piecewise_flag": (i, [vp]), "hg_last_forward_kernel": (i, [vp]), "hg_set_option": (i, [vp, C.c_char_p, i]), "hg_redone_frames": (C.c_long, [vp]), "hg_layout_walks": (C.c_long, [vp]), "hg_xcc_count": (i, [vp]),
        "hg_selftest_division": (i, [vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
        "hg_projective_plain_range": (i, [f64p, Geom]), "hg_affine_one_fma_form": (i, [f32p, Geom]), "hg_forward_tiles_admissible": (i, [i, f64p, i, i, Geom]),
        "hg_set_timing": (i, [vp, i]), "hg_last_kernel_ms": (i, [vp, f32p]),
        "hg_kernel_ms_stats": (i, [vp, f64p, C.POINTER(i)]),
        "hg_warp_forward_geometric": (i, [vp, i, f64p, Geom, u8p]),
        "hg_warp_forward_piecewise": (i, [vp, f32p, i, i, Geom, u8p]),
        "hg_warp_forward_geometric_device": (i, [vp, i, f64p, Geom, vp]),
        "hg_warp_forward_geometric_batch_device": (i, [vp, i, f64p, C.POINTER(Geom), C.POINTER(sz), i, vp]),
        "hg_warp_forward_piecewise_device": (i, [vp, f32p, i, i, Geom, vp]),
        "hg_warp_forward_piecewise_batch_device": (i, [vp, f32p, i, i, C.POINTER(Geom), C.POINTER(sz), i, vp]),
        "hg_upload_on_copy_stream": (i, [vp, vp, vp, sz]), "hg_fence_copies": (i, [vp]), "hg_download_behind_warps": (i, [vp, vp, vp, sz]), "hg_fence_downloads": (i, [vp]),
        "hg_solve_affine_triangles": (i, [f32p, f32p, i, C.POINTER(C.c_uint32), i, f32p]),
        "hg_warp_inverse_piecewise_state": (i, [vp, f32p, i, C.POINTER(TriMapDef), i, i, Geom, u8p]),
        "hg_warp_forward_piecewise_state": (i, [vp, f32p, i, C.POINTER(TriMapDef), i, i, i, i, Geom, u8p]),
    }
    older = "HGWARP_LIB" in os.environ       # A/B tooling (tools/ab*.sh) loads an OLDER build beside the current one: it may lack newer symbols
    for name, (res, args) in sig.items():
        if older and not hasattr(L, name):
            continue
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _geoms(gs):
    arr = (Geom * len(gs))(*[Geom(*[int(v) for v in g]) for g in gs])
    return arr


def _check(code, ctx=None):
    if code != 0:
        msg = lib().hg_last_error(ctx)
        raise HgError(code, msg.decode() if msg else "?")


# ------------------------------------------------------------------ host-side solves (no GPU needed)

def solve_affine(src, dst):
    (_, s), (_, d) = _f32(src), _f32(dst)
    out = np.empty(6, np.float32)
    _check(lib().hg_solve_affine(s, d, out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def solve_affine_triangles(src_pts, dst_pts, tris):
    """affineMatrixFromTriangles per triangle of a mesh (:785-804): (T, 6) float32, on the host."""
    (s, sp), (d, dp) = _f32(src_pts), _f32(dst_pts)
    t = np.ascontiguousarray(tris, np.uint32)
    out = np.empty((t.size // 3, 6), np.float32)
    _check(lib().hg_solve_affine_triangles(sp, dp, min(s.size, d.size) // 2, t.ctypes.data_as(C.POINTER(C.c_uint32)), t.size // 3,
                                           out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def _map_def(points, tris, width, height, y_off):
    p, pp = _f32(points)
    t = np.ascontiguousarray(tris, np.uint32)
    return TriMapDef(pp, p.size // 2, t.ctypes.data_as(C.POINTER(C.c_uint32)), t.size // 3, int(width), int(height), int(y_off)), (p, t)


def invert_affine(m):
    _, p = _f32(m)
    out = np.empty(6, np.float32)
    _check(lib().hg_invert_affine(p, out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def solve_projective(src, dst):
    (_, s), (_, d) = _f32(src), _f32(dst)
    out = np.empty(8, np.float64)
    _check(lib().hg_solve_projective(s, d, out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def transform_limits(kind, m, w, h):
    _, p = _f64(m)
    out = np.empty(4, np.float64)
    _check(lib().hg_transform_limits(int(kind), p, float(w), float(h), out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def minmax_xy(pts):
    a, p = _f32(pts)
    out = np.empty(4, np.float64)
    _check(lib().hg_minmax_xy(p, a.size, out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def projective_plain_range(m, geom):
    """True if the projective kernel may use its shared-reciprocal division for this frame (host-side range proof)."""
    a = np.ascontiguousarray(m, np.float64)
    return bool(lib().hg_projective_plain_range(a.ctypes.data_as(C.POINTER(C.c_double)), Geom(*[int(v) for v in geom])))


def affine_one_fma_form(inv, geom):
    """1 where the piecewise row kernel may evaluate an inverse matrix's two sums with one fma each (hg_affine_one_fma_form)."""
    a = np.ascontiguousarray(inv, np.float32)
    return bool(lib().hg_affine_one_fma_form(a.ctypes.data_as(C.POINTER(C.c_float)), Geom(*[int(v) for v in geom])))


def forward_tiles_admissible(kind, m, w, h, geom):
    """0 = scatter + gather, 1 = k_fwd_tiles admissible, 2 = admissible with a trusted inverse (include/hgwarp.h)."""
    a = np.zeros(8, np.float64)
    mm = np.ascontiguousarray(m, np.float64).ravel()
    a[:mm.size] = mm
    return int(lib().hg_forward_tiles_admissible(int(kind), a.ctypes.data_as(C.POINTER(C.c_double)), int(w), int(h), Geom(*[int(v) for v in geom])))


def js_round(x):
    return lib().hg_js_round(float(x))


def triangulate(pts):
    """Delaunay triangles (uint32, 3 per triangle) of interleaved x,y points: where the reference calls Delaunator (:1216)."""
    a, p = _f32(pts)
    n = a.size // 2
    out = np.empty(max(2 * n, 1) * 3, np.uint32)
    cnt = C.c_int(0)
    _check(lib().hg_triangulate(p, n, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size // 3, C.byref(cnt)))
    return out[:3 * cnt.value].copy()


def pack_offsets(geoms):
    g = _geoms(geoms)
    offs = (C.c_size_t * len(geoms))()
    total = C.c_size_t(0)
    _check(lib().hg_pack_offsets(g, len(geoms), offs, C.byref(total)))
    return list(offs), total.value


def device_count():
    n = C.c_int(0)
    code = lib().hg_device_count(C.byref(n))
    return n.value if code == 0 else 0


# ------------------------------------------------------------------ context

class Context:
    """One hg_ctx (one GPU, one stream) == the device side of one Homography instance."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        self._n_pts = self._n_tris = 0
        L = lib()
        if stream is None:
            code = L.hg_create(int(device), C.byref(self._h))
        else:
            code = L.hg_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(self._h))
        if code != 0:
            self._h = C.c_void_p()
            _check(code)

    def close(self):
        if self._h:
            lib().hg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _c(self, code):
        _check(code, self._h)

    # ---- plumbing
    def sync(self):
        self._c(lib().hg_sync(self._h))

    def alloc(self, nbytes):
        p = C.c_void_p()
        self._c(lib().hg_device_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value

    def free(self, dptr):
        self._c(lib().hg_device_free(self._h, C.c_void_p(dptr)))

    def to_host(self, dptr, nbytes, offset=0):
        out = np.empty(int(nbytes), np.uint8)
        self._c(lib().hg_copy_to_host(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(dptr + offset), int(nbytes)))
        return out

    def download_behind_warps(self, out, dptr, offset=0):
        """D2H into the uint8 array `out` on the context's download stream, ordered behind the warps issued so far (hg_download_behind_warps);
        `out` must stay alive until fence_downloads()."""
        self._c(lib().hg_download_behind_warps(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(dptr + offset), int(out.nbytes)))

    def fence_downloads(self):
        self._c(lib().hg_fence_downloads(self._h))

    def to_device(self, dptr, arr, offset=0):
        a = np.ascontiguousarray(arr)
        self._c(lib().hg_copy_to_device(self._h, C.c_void_p(dptr + offset), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def set_timing(self, on=True):
        self._c(lib().hg_set_timing(self._h, 1 if on else 0))

    def last_kernel_ms(self):
        ms = C.c_float(0)
        self._c(lib().hg_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def last_forward_kernel(self):
        """1 = scatter + gather, 2 = k_fwd_tiles (see include/hgwarp.h)"""
        return lib().hg_last_forward_kernel(self._h)

    def last_piecewise_kernel(self):
        """0 none, 1 k_pw_rows (4-row groups), 2 k_pw_rows (1 row), 3 k_pw_patch, 4 k_pw_fused, 5 k_pw_tile."""
        return lib().hg_last_piecewise_kernel(self._h)

    def last_piecewise_variant(self):
        """Variant code of the instantiation that ran (include/hgwarp.h); 0 with an older library."""
        L = lib()
        return L.hg_last_piecewise_variant(self._h) if hasattr(L, "hg_last_piecewise_variant") else 0

    def last_piecewise_self(self):
        """1 if the row workgroups of the last fused run evaluated their own spans (no row lists), 0 if they read k_tri_spans' lists."""
        return lib().hg_last_piecewise_self(self._h)

    def last_piecewise_flag(self):
        """Status word of the last frame a fused run flagged (diagnostics; include/hgwarp.h)."""
        return lib().hg_last_piecewise_flag(self._h)

    def redone_frames(self):
        """Frames redone through the materialised map since the context was created."""
        return lib().hg_redone_frames(self._h)

    def layout_walks(self):
        """Host walks over the triangles of a frame set (layout estimates) since the context was created."""
        return lib().hg_layout_walks(self._h)

    def xcc_count(self):
        """XCCs the warp kernels' block id -> row band mapping assumes (read from the device at creation, or option 'xcc')."""
        return lib().hg_xcc_count(self._h)

    def set_option(self, key, value):
        """Layout knobs of the piecewise fast path ("min_row_groups", "patch"); results never depend on them."""
        self._c(lib().hg_set_option(self._h, key.encode(), int(value)))

    def selftest_division(self, samples, seed=1):
        """Mismatches between the shared-reciprocal division of the projective kernel and IEEE division (must be 0)."""
        bad = C.c_uint64(0)
        self._c(lib().hg_selftest_division(self._h, int(samples), int(seed), C.byref(bad)))
        return bad.value

    def kernel_ms_stats(self):
        """(total ms, launches) of the dominant kernel since set_timing(True)."""
        tot, n = C.c_double(0), C.c_int(0)
        self._c(lib().hg_kernel_ms_stats(self._h, C.byref(tot), C.byref(n)))
        return tot.value, n.value

    # ---- image
    def set_image(self, rgba):
        a = np.ascontiguousarray(rgba, dtype=np.uint8)
        h, w = a.shape[:2]
        self._c(lib().hg_set_image(self._h, a.ctypes.data_as(C.POINTER(C.c_uint8)), w, h))

    def set_image_device(self, dptr, w, h):
        self._c(lib().hg_set_image_device(self._h, C.c_void_p(int(dptr)), int(w), int(h)))

    def set_images_device(self, dptr, w, h, n_images, stride_bytes):
        self._c(lib().hg_set_images_device(self._h, C.c_void_p(int(dptr)), int(w), int(h), int(n_images), int(stride_bytes)))

    # ---- affine / projective
    def warp_inverse_geometric(self, kind, m, geom):
        _, p = _f64(m)
        g = Geom(*[int(v) for v in geom])
        out = np.zeros((max(g.obj_h, 0), max(g.obj_w, 0), 4), np.uint8)
        self._c(lib().hg_warp_inverse_geometric(self._h, int(kind), p, g, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def warp_inverse_geometric_device(self, kind, m, geom, d_out):
        """One affine / projective frame into device memory (asynchronous; sync() settles it)."""
        a, p = _f64(m)
        assert a.size >= (6 if int(kind) == 0 else 8)
        self._c(lib().hg_warp_inverse_geometric_device(self._h, int(kind), p, Geom(*[int(v) for v in geom]), C.c_void_p(int(d_out))))

    def warp_forward_geometric(self, kind, m, geom):
        _, p = _f64(m)
        g = Geom(*[int(v) for v in geom])
        out = np.zeros((max(g.obj_h, 0), max(g.obj_w, 0), 4), np.uint8)
        self._c(lib().hg_warp_forward_geometric(self._h, int(kind), p, g, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def warp_forward_piecewise(self, dst_pts, max_src_x, max_src_y, geom):
        d, dp = _f32(dst_pts)
        assert d.size == 2 * self._n_pts, "one x,y pair per mesh point"
        g = Geom(*[int(v) for v in geom])
        out = np.zeros((max(g.obj_h, 0), max(g.obj_w, 0), 4), np.uint8)
        self._c(lib().hg_warp_forward_piecewise(self._h, dp, int(max_src_x), int(max_src_y), g, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    # reference-state forms: the cached matrices as they stood + the definition of the map the shared field held
    def warp_inverse_piecewise_state(self, fwd_mats, dst_pts, tris, min_src_x, min_src_y, geom):
        m, mp = _f32(fwd_mats)
        g = Geom(*[int(v) for v in geom])
        d, keep = _map_def(dst_pts, tris, g.obj_w, g.obj_h, g.y_off)
        out = np.zeros((max(g.obj_h, 0), max(g.obj_w, 0), 4), np.uint8)
        self._c(lib().hg_warp_inverse_piecewise_state(self._h, mp, m.size // 6, C.byref(d), int(min_src_x), int(min_src_y), g, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def warp_forward_piecewise_state(self, fwd_mats, map_pts, map_tris, map_w, map_h, map_y_off, min_src_x, min_src_y, max_src_x, max_src_y, geom):
        m, mp = _f32(fwd_mats)
        g = Geom(*[int(v) for v in geom])
        d, keep = _map_def(map_pts, map_tris, map_w, map_h, map_y_off)
        out = np.zeros((max(g.obj_h, 0), max(g.obj_w, 0), 4), np.uint8)
        self._c(lib().hg_warp_forward_piecewise_state(self._h, mp, m.size // 6, C.byref(d), int(min_src_x), int(min_src_y), int(max_src_x), int(max_src_y), g,
                                                      out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def warp_forward_geometric_batch_device(self, kind, mats, geoms, offsets, d_out):
        m, p = _f64(mats)
        assert m.size == 8 * len(geoms)
        offs = (C.c_size_t * len(geoms))(*offsets) if offsets is not None else None
        self._c(lib().hg_warp_forward_geometric_batch_device(self._h, int(kind), p, _geoms(geoms), offs, len(geoms), C.c_void_p(int(d_out))))

    def warp_forward_piecewise_batch_device(self, dst_pts, max_src_x, max_src_y, geoms, offsets, d_out):
        d, dp = _f32(dst_pts)
        assert d.size == 2 * self._n_pts * len(geoms), "frames x mesh points x,y pairs"
        offs = (C.c_size_t * len(geoms))(*offsets) if offsets is not None else None
        self._c(lib().hg_warp_forward_piecewise_batch_device(self._h, dp, int(max_src_x), int(max_src_y), _geoms(geoms), offs, len(geoms), C.c_void_p(int(d_out))))

    def geometric_set_frames(self, kind, mats, geoms, offsets=None):
        m, p = _f64(mats)
        assert m.size == 8 * len(geoms)
        offs = (C.c_size_t * len(geoms))(*offsets) if offsets is not None else None
        self._c(lib().hg_geometric_set_frames(self._h, int(kind), p, _geoms(geoms), offs, len(geoms)))

    def geometric_set_frames_points(self, kind, from_pts, to_pts, geoms, offsets=None):
        """Frames as point sets: the matrices (from -> to) are solved on the device at every warp, like the reference's :994."""
        per = 6 if int(kind) == 0 else 8
        (a, ap), (b, bp) = _f32(from_pts), _f32(to_pts)
        assert a.size == per * len(geoms) and b.size == per * len(geoms)
        offs = (C.c_size_t * len(geoms))(*offsets) if offsets is not None else None
        self._c(lib().hg_geometric_set_frames_points(self._h, int(kind), ap, bp, _geoms(geoms), offs, len(geoms)))

    def geometric_points_args(self, kind, from_pts, to_pts, geoms, offsets=None):
        """The ctypes arguments of geometric_set_frames_points, built once (bench.py --points fresh)."""
        per = 6 if int(kind) == 0 else 8
        (a, ap), (b, bp) = _f32(from_pts), _f32(to_pts)
        assert a.size == per * len(geoms) and b.size == per * len(geoms)
        offs = (C.c_size_t * len(geoms))(*offsets) if offsets is not None else None
        return (a, b, int(kind), ap, bp, _geoms(geoms), offs, len(geoms))

    def geometric_set_frames_points_prepared(self, args):
        self._c(lib().hg_geometric_set_frames_points(self._h, args[2], args[3], args[4], args[5], args[6], args[7]))

    def get_geometric_matrices(self, n_frames):
        out = np.empty((int(n_frames), 8), np.float64)
        self._c(lib().hg_get_geometric_matrices(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), int(n_frames)))
        return out

    def warp_inverse_geometric_frames_device(self, d_out):
        self._c(lib().hg_warp_inverse_geometric_frames_device(self._h, C.c_void_p(int(d_out))))

    # ---- piecewise
    def piecewise_set_mesh(self, src_pts, tris, min_src_x, min_src_y):
        s, sp = _f32(src_pts)
        t = np.ascontiguousarray(tris, dtype=np.uint32)
        self._c(lib().hg_piecewise_set_mesh(self._h, sp, s.size // 2, t.ctypes.data_as(C.POINTER(C.c_uint32)), t.size // 3,
                                            int(min_src_x), int(min_src_y)))
        self._n_pts, self._n_tris = s.size // 2, t.size // 3      # the C ABI takes plain pointers: sizes are checked on this side

    def piecewise_prepare(self, dst_pts, geom):
        d, dp = _f32(dst_pts)
        assert d.size == 2 * self._n_pts, "one x,y pair per mesh point"
        self._geom = Geom(*[int(v) for v in geom])
        self._c(lib().hg_piecewise_prepare(self._h, dp, self._geom))

    def _out_array(self):
        g = self._geom
        return np.zeros((max(g.obj_h, 0), max(g.obj_w, 0), 4), np.uint8)

    def warp_inverse_piecewise(self):
        out = self._out_array()
        self._c(lib().hg_warp_inverse_piecewise(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def warp_inverse_piecewise_via_map(self):
        out = self._out_array()
        self._c(lib().hg_warp_inverse_piecewise_via_map(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def get_tri_map(self, fused=False):
        g = self._geom
        m = np.empty(max(g.obj_w, 0) * max(g.obj_h, 0), np.int16)
        fn = lib().hg_get_tri_map_fused if fused else lib().hg_get_tri_map
        self._c(fn(self._h, m.ctypes.data_as(C.POINTER(C.c_int16)), m.size))
        return m

    def get_matrices(self, n_tris):
        assert n_tris == self._n_tris, "the library fills n_triangles x 6 floats"
        fwd = np.empty((n_tris, 6), np.float32)
        inv = np.empty((n_tris, 6), np.float32)
        self._c(lib().hg_get_matrices(self._h, fwd.ctypes.data_as(C.POINTER(C.c_float)), inv.ctypes.data_as(C.POINTER(C.c_float))))
        return fwd, inv

    def piecewise_set_frames(self, dst_pts, geoms, offsets=None):
        d, dp = _f32(dst_pts)
        assert d.size == 2 * self._n_pts * len(geoms), "frames x mesh points x,y pairs"
        offs = (C.c_size_t * len(geoms))(*offsets) if offsets is not None else None
        self._c(lib().hg_piecewise_set_frames(self._h, dp, _geoms(geoms), offs, len(geoms)))

    def frame_set_args(self, dst_pts, geoms, offsets=None):
        """The ctypes arguments of piecewise_set_frames, built once: a caller that uploads one of a few point sets per step
        (bench.py --points fresh) then pays only the C call.  Returns an opaque tuple for piecewise_set_frames_prepared."""
        d, dp = _f32(dst_pts)
        assert d.size == 2 * self._n_pts * len(geoms), "frames x mesh points x,y pairs"
        offs = (C.c_size_t * len(geoms))(*offsets) if offsets is not None else None
        return (d, dp, _geoms(geoms), offs, len(geoms))

    def piecewise_set_frames_prepared(self, args):
        self._c(lib().hg_piecewise_set_frames(self._h, args[1], args[2], args[3], args[4]))

    def warp_inverse_piecewise_frames_device(self, d_out):
        self._c(lib().hg_warp_inverse_piecewise_frames_device(self._h, C.c_void_p(int(d_out))))


# ------------------------------------------------------------------ several GPUs, one host thread (hg_multi_*)

def multi_plan_fanout(access):
    """Copy plan of the shared-source fan-out (pure function, no GPU): access = G x G array, access[p, q] != 0 when device q copies
    straight out of device p's memory.  Returns a list of (dst, src or -1 = host buffer, slice, phase)."""
    a = np.ascontiguousarray(access, np.uint8)
    G = a.shape[0]
    assert a.shape == (G, G)
    n = lib().hg_multi_plan_fanout(G, a.ctypes.data, None, 0)
    if n < 0:
        raise RuntimeError("hg_multi_plan_fanout: bad arguments")
    ops = np.zeros((max(n, 1), 4), np.int32)
    lib().hg_multi_plan_fanout(G, a.ctypes.data, ops.ctypes.data, n)
    return [tuple(int(v) for v in ops[k]) for k in range(n)]


def multi_partition(n_frames, n_devices, index):
    """(first, count): the contiguous block of frames device `index` of `n_devices` warps (pure function, no GPU)."""
    first, count = C.c_int(0), C.c_int(0)
    _check(lib().hg_multi_partition(int(n_frames), int(n_devices), int(index), C.byref(first), C.byref(count)))
    return first.value, count.value


class PinnedBuffer:
    """Page-locked host memory from hg_host_alloc, exposed as a numpy uint8 array."""

    def __init__(self, nbytes):
        self._p = C.c_void_p()
        _check(lib().hg_host_alloc(int(nbytes), C.byref(self._p)))
        self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_uint8)), shape=(int(nbytes),))

    @property
    def ptr(self):
        return self._p.value

    def free(self):
        if self._p:
            self.array = None
            lib().hg_host_free(self._p)
            self._p = C.c_void_p()


class Multi:
    """hg_multi: the batch caller loop spread over several devices of one node from ONE host thread."""

    def __init__(self, devices):
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._h = C.c_void_p()
        code = lib().hg_multi_create(ids, len(devices), C.byref(self._h))
        if code != 0:
            self._h = C.c_void_p()
            msg = lib().hg_multi_last_error(None)
            raise HgError(code, msg.decode() if msg else "?")
        self._n_pts = 0
        self._geoms = []

    def _c(self, code):
        if code != 0:
            msg = lib().hg_multi_last_error(self._h)
            raise HgError(code, msg.decode() if msg else "?")

    def peer_note(self):
        """'' when every pair of distinct devices has peer access, else a text naming the pairs without it."""
        msg = lib().hg_multi_peer_note(self._h)
        return msg.decode() if msg else ""

    def peer_access(self, from_index, to_index):
        return lib().hg_multi_peer_access(self._h, int(from_index), int(to_index))

    def close(self):
        if self._h:
            lib().hg_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_count(self):
        return lib().hg_multi_device_count(self._h)

    def set_option(self, key, value):
        for k in range(self.device_count()):
            ctx = lib().hg_multi_ctx(self._h, k)
            _check(lib().hg_set_option(C.c_void_p(ctx), key.encode(), int(value)), C.c_void_p(ctx))

    def set_image(self, rgba):
        a = np.ascontiguousarray(rgba, dtype=np.uint8)
        h, w = a.shape[:2]
        self._c(lib().hg_multi_set_image(self._h, a.ctypes.data_as(C.POINTER(C.c_uint8)), w, h))

    def piecewise_set_mesh(self, src_pts, tris, min_src_x, min_src_y):
        s, sp = _f32(src_pts)
        t = np.ascontiguousarray(tris, dtype=np.uint32)
        self._c(lib().hg_multi_piecewise_set_mesh(self._h, sp, s.size // 2, t.ctypes.data_as(C.POINTER(C.c_uint32)), t.size // 3, int(min_src_x), int(min_src_y)))
        self._n_pts = s.size // 2

    def warp_piecewise_batch(self, dst_pts, geoms, out_ptrs=None):
        """out_ptrs: None (frames stay on their devices) or one host address per frame."""
        d, dp = _f32(dst_pts)
        assert d.size == 2 * self._n_pts * len(geoms)
        ptrs = (C.c_void_p * len(geoms))(*[C.c_void_p(int(p)) for p in out_ptrs]) if out_ptrs is not None else None
        self._geoms = [tuple(int(v) for v in g) for g in geoms]
        self._c(lib().hg_multi_warp_piecewise_batch(self._h, dp, _geoms(geoms), len(geoms), ptrs))

    @staticmethod
    def _image_ptrs(images, n):
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        assert len(imgs) == n and all(im.shape == imgs[0].shape and im.ndim == 3 and im.shape[2] == 4 for im in imgs), "one H x W x 4 image per frame"
        return imgs, (C.c_void_p * n)(*[C.c_void_p(im.ctypes.data) for im in imgs]), imgs[0].shape[1], imgs[0].shape[0]

    def warp_piecewise_batch_images(self, dst_pts, geoms, images, out_ptrs=None):
        """One source per frame (the video loop): images[f] is frame f's H x W x 4 uint8 source; every device uploads its own block."""
        d, dp = _f32(dst_pts)
        assert d.size == 2 * self._n_pts * len(geoms)
        keep, iptrs, w, h = self._image_ptrs(images, len(geoms))
        ptrs = (C.c_void_p * len(geoms))(*[C.c_void_p(int(p)) for p in out_ptrs]) if out_ptrs is not None else None
        self._geoms = [tuple(int(v) for v in g) for g in geoms]
        self._c(lib().hg_multi_warp_piecewise_batch_images(self._h, dp, _geoms(geoms), len(geoms), iptrs, w, h, ptrs))

    def warp_geometric_batch_images(self, kind, from_pts, to_pts, geoms, images, out_ptrs=None):
        per = 6 if int(kind) == 0 else 8
        (a, ap), (b, bp) = _f32(from_pts), _f32(to_pts)
        assert a.size == per * len(geoms) and b.size == per * len(geoms)
        keep, iptrs, w, h = self._image_ptrs(images, len(geoms))
        ptrs = (C.c_void_p * len(geoms))(*[C.c_void_p(int(p)) for p in out_ptrs]) if out_ptrs is not None else None
        self._geoms = [tuple(int(v) for v in g) for g in geoms]
        self._c(lib().hg_multi_warp_geometric_batch_images(self._h, int(kind), ap, bp, _geoms(geoms), len(geoms), iptrs, w, h, ptrs))

    def warp_geometric_batch(self, kind, from_pts, to_pts, geoms, out_ptrs=None):
        per = 6 if int(kind) == 0 else 8
        (a, ap), (b, bp) = _f32(from_pts), _f32(to_pts)
        assert a.size == per * len(geoms) and b.size == per * len(geoms)
        ptrs = (C.c_void_p * len(geoms))(*[C.c_void_p(int(p)) for p in out_ptrs]) if out_ptrs is not None else None
        self._geoms = [tuple(int(v) for v in g) for g in geoms]
        self._c(lib().hg_multi_warp_geometric_batch(self._h, int(kind), ap, bp, _geoms(geoms), len(geoms), ptrs))

    def frame(self, f):
        """(device index, device pointer, bytes) of frame f of the last batch."""
        dev, ptr, n = C.c_int(0), C.c_void_p(), C.c_size_t(0)
        self._c(lib().hg_multi_frame(self._h, int(f), C.byref(dev), C.byref(ptr), C.byref(n)))
        return dev.value, ptr.value, n.value

    def frame_to_host(self, f):
        dev, ptr, n = self.frame(f)
        out = np.empty(n, np.uint8)
        ctx = C.c_void_p(lib().hg_multi_ctx(self._h, dev))
        _check(lib().hg_copy_to_host(ctx, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), n), ctx)
        g = self._geoms[f]
        return out.reshape(max(g[3], 0), max(g[2], 0), 4)
