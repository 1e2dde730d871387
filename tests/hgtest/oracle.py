"""ctypes binding of oracle/libhgoracle.so (the CPU checker; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(ORACLE_DIR, "libhgoracle.so")
    src = os.path.join(ORACLE_DIR, "hg_oracle.c")
    if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        build()
    L = C.CDLL(so)
    L.hgo_js_round.restype = C.c_double
    L.hgo_js_round.argtypes = [C.c_double]
    L.hgo_affine_from_triangles.argtypes = [_f32p, _f32p, _f32p]
    L.hgo_inverse_affine.argtypes = [_f32p, _f32p]
    L.hgo_projective_from_squares.argtypes = [_f32p, _f32p, _f64p]
    L.hgo_fill_triangle.argtypes = [_f32p, C.c_double, C.c_double, C.c_double, _i16p, C.c_int64]
    L.hgo_build_tri_map.argtypes = [_f32p, _u32p, C.c_int, C.c_double, C.c_double, _i16p, C.c_int64]
    L.hgo_piecewise_matrices.argtypes = [_f32p, _f32p, _u32p, C.c_int, _f32p]
    L.hgo_transform_limits.argtypes = [C.c_int, _f64p, C.c_double, C.c_double, _f64p]
    L.hgo_minmax_xy.argtypes = [_f32p, C.c_int, _f64p]
    L.hgo_warp_inverse_geometric.argtypes = [C.c_int, _f64p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p]
    L.hgo_warp_inverse_piecewise_loop.argtypes = [_i16p, _f32p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_int, C.c_int, C.c_int, C.c_int, _u8p]
    L.hgo_warp_inverse_piecewise.argtypes = [_f32p, _f32p, _u32p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hgo_warp_forward_geometric.argtypes = [C.c_int, _f64p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p]
    L.hgo_warp_forward_piecewise.argtypes = [_i16p, _f32p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_int, _u8p]
    L.hgo_lcg_image.argtypes = [_u8p, C.c_size_t, C.c_uint32]
    _LIB = L
    return L


# ---------------------------------------------------------------- numpy-friendly wrappers

def js_round(x):
    return lib().hgo_js_round(float(x))


def affine_from_triangles(src, dst):
    out = np.empty(6, np.float32)
    lib().hgo_affine_from_triangles(np.ascontiguousarray(src, np.float32), np.ascontiguousarray(dst, np.float32), out)
    return out


def inverse_affine(m):
    out = np.empty(6, np.float32)
    lib().hgo_inverse_affine(np.ascontiguousarray(m, np.float32), out)
    return out


def projective_from_squares(src, dst):
    out = np.empty(8, np.float64)
    lib().hgo_projective_from_squares(np.ascontiguousarray(src, np.float32), np.ascontiguousarray(dst, np.float32), out)
    return out


def fill_triangle(tri, idx, width, yoff, map_):
    lib().hgo_fill_triangle(np.ascontiguousarray(tri, np.float32), float(idx), float(width), float(yoff), map_, map_.size)
    return map_


def build_tri_map(points, tris, width, yoff, n_cells):
    m = np.empty(max(int(n_cells), 0), np.int16)
    tris = np.ascontiguousarray(tris, np.uint32)
    lib().hgo_build_tri_map(np.ascontiguousarray(points, np.float32), tris, tris.size // 3, float(width), float(yoff), m, m.size)
    return m


def piecewise_matrices(src_pts, dst_pts, tris):
    tris = np.ascontiguousarray(tris, np.uint32)
    fwd = np.empty((tris.size // 3, 6), np.float32)
    lib().hgo_piecewise_matrices(np.ascontiguousarray(src_pts, np.float32), np.ascontiguousarray(dst_pts, np.float32),
                                 tris, tris.size // 3, fwd)
    return fwd


def transform_limits(kind, m, w, h):
    out = np.empty(4, np.float64)
    lib().hgo_transform_limits(int(kind), np.ascontiguousarray(m, np.float64), float(w), float(h), out)
    return out


def minmax_xy(pts):
    pts = np.ascontiguousarray(pts, np.float32)
    out = np.empty(4, np.float64)
    lib().hgo_minmax_xy(pts, pts.size, out)
    return out


def warp_inverse_geometric(kind, m, image, xoff, yoff, objw, objh):
    image = np.ascontiguousarray(image, np.uint8)
    H, W = image.shape[:2]
    out = np.zeros((max(objh, 0), max(objw, 0), 4), np.uint8)
    lib().hgo_warp_inverse_geometric(int(kind), np.ascontiguousarray(m, np.float64), image, W, H, xoff, yoff, objw, objh, out)
    return out


def warp_inverse_piecewise(src_pts, dst_pts, tris, image, min_src_x, min_src_y, xoff, yoff, objw, objh, taps=False):
    image = np.ascontiguousarray(image, np.uint8)
    H, W = image.shape[:2]
    tris = np.ascontiguousarray(tris, np.uint32)
    T = tris.size // 3
    out = np.zeros((max(objh, 0), max(objw, 0), 4), np.uint8)
    map_ = np.empty(max(objw, 0) * max(objh, 0), np.int16)
    fwd = np.empty((T, 6), np.float32)
    inv = np.empty((T, 6), np.float32)
    lib().hgo_warp_inverse_piecewise(np.ascontiguousarray(src_pts, np.float32), np.ascontiguousarray(dst_pts, np.float32),
                                     tris, T, image, W, H, min_src_x, min_src_y, xoff, yoff, objw, objh, out,
                                     map_.ctypes.data, fwd.ctypes.data, inv.ctypes.data)
    return (out, map_, fwd, inv) if taps else out


def warp_inverse_piecewise_loop(map16, inv, image, min_src_x, min_src_y, xoff, yoff, objw, objh):
    """The pixel loop :1042-1056 alone, over a given Int16 map and the INVERSE matrices it indexes ((T, 6) float32)."""
    image = np.ascontiguousarray(image, np.uint8)
    H, W = image.shape[:2]
    inv = np.ascontiguousarray(inv, np.float32)
    out = np.zeros((max(objh, 0), max(objw, 0), 4), np.uint8)
    lib().hgo_warp_inverse_piecewise_loop(np.ascontiguousarray(map16, np.int16), inv, inv.size // 6, image, W, H, min_src_x, min_src_y,
                                          xoff, yoff, objw, objh, out)
    return out


def warp_forward_geometric(kind, m, image, xoff, yoff, objw, objh):
    image = np.ascontiguousarray(image, np.uint8)
    H, W = image.shape[:2]
    out = np.zeros((max(objh, 0), max(objw, 0), 4), np.uint8)
    lib().hgo_warp_forward_geometric(int(kind), np.ascontiguousarray(m, np.float64), image, W, H, xoff, yoff, objw, objh, out)
    return out


def warp_forward_piecewise(fwd_map, fwd, image, min_src_x, min_src_y, max_src_x, max_src_y, xoff, yoff, objw, objh):
    image = np.ascontiguousarray(image, np.uint8)
    H, W = image.shape[:2]
    fwd = np.ascontiguousarray(fwd, np.float32)
    out = np.zeros((max(objh, 0), max(objw, 0), 4), np.uint8)
    lib().hgo_warp_forward_piecewise(np.ascontiguousarray(fwd_map, np.int16), fwd, fwd.size // 6, image, W, H,
                                     min_src_x, min_src_y, max_src_x, max_src_y, xoff, yoff, objw, objh, out)
    return out


def lcg_image(w, h, seed):
    out = np.empty((h, w, 4), np.uint8)
    lib().hgo_lcg_image(out, out.size, seed)
    return out
