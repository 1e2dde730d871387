/*
 * hgwarp_napi.c -- thin N-API addon over the C ABI (include/hgwarp.h) for the drop-in JS class js/Homography.mjs.
 *
 * It only unwraps TypedArrays, calls libhgwarp.so on the main JS thread (the reference's warp() is synchronous too)
 * and creates the result arrays; failures are thrown as bare strings like the reference does (`throw("...")`).
 * Built with plain gcc against /usr/include/node (Makefile target lib/hgwarp.node); no node-gyp, no C++.
 */
#include <node_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hgwarp.h"

#define NAPI_OK(call) do { if ((call) != napi_ok) { napi_throw_error(env, NULL, "hgwarp: N-API call failed: " #call); return NULL; } } while (0)

static napi_value throw_str(napi_env env, const char *msg)
{
    napi_value s;
    napi_create_string_utf8(env, msg, NAPI_AUTO_LENGTH, &s);
    napi_throw(env, s);                       /* bare string, like the reference's throw("...") */
    return NULL;
}

static napi_value throw_hg(napi_env env, hg_ctx *ctx, const char *what, int code)
{
    char buf[512];
    snprintf(buf, sizeof buf, "hgwarp %s failed (%d): %s", what, code, hg_last_error(ctx));
    return throw_str(env, buf);
}

#define HG_CALL(ctx, what, call) do { int rc_ = (call); if (rc_ != HG_OK) return throw_hg(env, (ctx), (what), rc_); } while (0)

/* d_batch: device buffer the frames of warpInversePiecewiseBatch are produced in (kept between calls, grown as needed) */
/* n_pts / n_tris: the mesh last set, so that point-set and matrix buffers can be checked before the C ABI reads / fills them */
typedef struct { hg_ctx *ctx; int obj_w, obj_h; void *d_batch; size_t d_batch_cap; size_t n_pts, n_tris; } handle_t;

static void release_ctx(handle_t *h)
{
    if (h->ctx) {
        if (h->d_batch) hg_device_free(h->ctx, h->d_batch);
        hg_destroy(h->ctx);
    }
    h->ctx = NULL; h->d_batch = NULL; h->d_batch_cap = 0;
}

static void handle_finalize(napi_env env, void *data, void *hint)
{
    (void)env; (void)hint;
    handle_t *h = (handle_t *)data;
    if (h) { release_ctx(h); free(h); }
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv)
{
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) { throw_str(env, "hgwarp: wrong number of arguments"); return 0; }
    return 1;
}

static handle_t *get_handle(napi_env env, napi_value v)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((handle_t *)p)->ctx) { throw_str(env, "hgwarp: invalid or destroyed context handle"); return NULL; }
    return (handle_t *)p;
}

/* typed array of an exact element type; returns data pointer and element count */
static void *get_typed(napi_env env, napi_value v, napi_typedarray_type want, size_t *len, const char *name)
{
    bool is = false;
    napi_typedarray_type t; size_t n = 0; void *data = NULL; napi_value ab; size_t off = 0;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is || napi_get_typedarray_info(env, v, &t, &n, &data, &ab, &off) != napi_ok ||
        (t != want && !(want == napi_uint8_clamped_array && t == napi_uint8_array))) {
        char buf[160]; snprintf(buf, sizeof buf, "hgwarp: argument '%s' has the wrong TypedArray type", name);
        throw_str(env, buf); return NULL;
    }
    *len = n;
    return data ? data : (void *)"";          /* zero-length arrays may report a NULL pointer */
}

static int get_i32(napi_env env, napi_value v, int *out)
{
    double d;
    if (napi_get_value_double(env, v, &d) != napi_ok) { throw_str(env, "hgwarp: expected a number"); return 0; }
    *out = (int)d;
    return 1;
}

static napi_value make_typed(napi_env env, napi_typedarray_type t, size_t n, size_t elem, void **data)
{
    napi_value ab, ta;
    NAPI_OK(napi_create_arraybuffer(env, n * elem, data, &ab));
    NAPI_OK(napi_create_typedarray(env, t, n, ab, 0, &ta));
    return ta;
}

/* RGBA outputs.  Default: a fresh, V8-owned Uint8ClampedArray per call like the reference (:991, :1040).  If the caller
 * passes its own Uint8ClampedArray of at least `bytes` bytes as the optional last argument, the frame is written there
 * and a view of exactly `bytes` bytes over the same memory is returned: no 34 MB allocation, first-touch page faults and
 * garbage per 4K frame (that, not PCIe, is most of the host time of a frame on this platform).  Measured and rejected:
 * pooled external ArrayBuffers -- Node 12 runs their finalizers only from the event loop, so a synchronous warp() loop
 * would hold every frame it ever produced. */
static napi_value make_pixels(napi_env env, size_t bytes, napi_value reuse, int have_reuse, void **data)
{
    if (have_reuse) {
        bool is = false;
        napi_typedarray_type t; size_t n = 0, off = 0; void *p = NULL; napi_value ab, ta;
        if (napi_is_typedarray(env, reuse, &is) == napi_ok && is &&
            napi_get_typedarray_info(env, reuse, &t, &n, &p, &ab, &off) == napi_ok &&
            (t == napi_uint8_clamped_array || t == napi_uint8_array) && n >= bytes && p) {
            NAPI_OK(napi_create_typedarray(env, napi_uint8_clamped_array, bytes, ab, off, &ta));
            *data = p;
            return ta;
        }
    }
    return make_typed(env, napi_uint8_clamped_array, bytes, 1, data);
}

/* optional trailing argument: argv[want] if present and not undefined/null */
static int get_args_opt(napi_env env, napi_callback_info info, size_t want, napi_value *argv, int *have_opt)
{
    size_t argc = want + 1;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) { throw_str(env, "hgwarp: wrong number of arguments"); return 0; }
    *have_opt = 0;
    if (argc > want) {
        napi_valuetype vt;
        if (napi_typeof(env, argv[want], &vt) == napi_ok && vt != napi_undefined && vt != napi_null) *have_opt = 1;
    }
    return 1;
}

/* ---------------------------------------------------------------- context */
static napi_value fn_create(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    int dev = 0;
    if (!get_i32(env, a[0], &dev)) return NULL;
    handle_t *h = (handle_t *)calloc(1, sizeof *h);
    int rc = hg_create(dev, &h->ctx);
    if (rc != HG_OK) { free(h); return throw_hg(env, NULL, "hg_create", rc); }
    napi_value ext;
    NAPI_OK(napi_create_external(env, h, handle_finalize, NULL, &ext));
    return ext;
}

static napi_value fn_destroy(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    void *p = NULL;
    if (napi_get_value_external(env, a[0], &p) == napi_ok && p) release_ctx((handle_t *)p);
    return NULL;
}

static napi_value fn_device_count(napi_env env, napi_callback_info info)
{
    (void)info;
    int n = 0;
    hg_device_count(&n);
    napi_value v;
    NAPI_OK(napi_create_int32(env, n, &v));
    return v;
}

/* ---------------------------------------------------------------- host solves */
static napi_value fn_solve_affine(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    size_t n0, n1;
    float *s = (float *)get_typed(env, a[0], napi_float32_array, &n0, "src"); if (!s) return NULL;
    float *d = (float *)get_typed(env, a[1], napi_float32_array, &n1, "dst"); if (!d) return NULL;
    if (n0 < 6 || n1 < 6) return throw_str(env, "hgwarp: solveAffine needs 6 values per point set");
    void *out; napi_value r = make_typed(env, napi_float32_array, 6, 4, &out); if (!r) return NULL;
    HG_CALL(NULL, "hg_solve_affine", hg_solve_affine(s, d, (float *)out));
    return r;
}

static napi_value fn_invert_affine(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    size_t n;
    float *m = (float *)get_typed(env, a[0], napi_float32_array, &n, "matrix"); if (!m) return NULL;
    if (n < 6) return throw_str(env, "hgwarp: invertAffine needs 6 values");
    void *out; napi_value r = make_typed(env, napi_float32_array, 6, 4, &out); if (!r) return NULL;
    HG_CALL(NULL, "hg_invert_affine", hg_invert_affine(m, (float *)out));
    return r;
}

static napi_value fn_solve_projective(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    size_t n0, n1;
    float *s = (float *)get_typed(env, a[0], napi_float32_array, &n0, "src"); if (!s) return NULL;
    float *d = (float *)get_typed(env, a[1], napi_float32_array, &n1, "dst"); if (!d) return NULL;
    if (n0 < 8 || n1 < 8) return throw_str(env, "hgwarp: solveProjective needs 8 values per point set");
    void *out; napi_value r = make_typed(env, napi_float64_array, 8, 8, &out); if (!r) return NULL;
    HG_CALL(NULL, "hg_solve_projective", hg_solve_projective(s, d, (double *)out));
    return r;
}

static napi_value fn_transform_limits(napi_env env, napi_callback_info info)
{
    napi_value a[4];
    if (!get_args(env, info, 4, a)) return NULL;
    int kind; size_t n; double w, h;
    if (!get_i32(env, a[0], &kind)) return NULL;
    double *m = (double *)get_typed(env, a[1], napi_float64_array, &n, "matrix"); if (!m) return NULL;
    if (n < (size_t)(kind == HG_AFFINE ? 6 : 8)) return throw_str(env, "hgwarp: matrix too short");
    if (napi_get_value_double(env, a[2], &w) != napi_ok || napi_get_value_double(env, a[3], &h) != napi_ok) return throw_str(env, "hgwarp: width/height must be numbers");
    void *out; napi_value r = make_typed(env, napi_float64_array, 4, 8, &out); if (!r) return NULL;
    HG_CALL(NULL, "hg_transform_limits", hg_transform_limits(kind, m, w, h, (double *)out));
    return r;
}

static napi_value fn_minmax_xy(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    size_t n;
    float *p = (float *)get_typed(env, a[0], napi_float32_array, &n, "points"); if (!p) return NULL;
    void *out; napi_value r = make_typed(env, napi_float64_array, 4, 8, &out); if (!r) return NULL;
    HG_CALL(NULL, "hg_minmax_xy", hg_minmax_xy(p, (int)n, (double *)out));
    return r;
}

static napi_value fn_triangulate(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    size_t n;
    float *p = (float *)get_typed(env, a[0], napi_float32_array, &n, "points"); if (!p) return NULL;
    int count = 0;
    HG_CALL(NULL, "hg_triangulate", hg_triangulate(p, (int)(n / 2), NULL, 0, &count));
    void *out; napi_value r = make_typed(env, napi_uint32_array, (size_t)count * 3, 4, &out); if (!r) return NULL;
    if (count) HG_CALL(NULL, "hg_triangulate", hg_triangulate(p, (int)(n / 2), (uint32_t *)out, count, &count));
    return r;
}

/* ---------------------------------------------------------------- image + warps */
static napi_value fn_set_image(napi_env env, napi_callback_info info)
{
    napi_value a[4];
    if (!get_args(env, info, 4, a)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    size_t n; int w, hh;
    uint8_t *px = (uint8_t *)get_typed(env, a[1], napi_uint8_clamped_array, &n, "image.data"); if (!px) return NULL;
    if (!get_i32(env, a[2], &w) || !get_i32(env, a[3], &hh)) return NULL;
    if (w <= 0 || hh <= 0 || n < (size_t)w * (size_t)hh * 4) return throw_str(env, "hgwarp: image.data is smaller than width*height*4");
    HG_CALL(h->ctx, "hg_set_image", hg_set_image(h->ctx, px, w, hh));
    return NULL;
}

static int get_geom(napi_env env, napi_value *a, hg_geom *g)
{
    int v[4];
    for (int i = 0; i < 4; i++) if (!get_i32(env, a[i], &v[i])) return 0;
    g->x_off = v[0]; g->y_off = v[1]; g->obj_w = v[2]; g->obj_h = v[3];
    return 1;
}

static napi_value fn_warp_inverse_geometric(napi_env env, napi_callback_info info)
{
    napi_value a[8]; int reuse = 0;
    if (!get_args_opt(env, info, 7, a, &reuse)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    int kind; size_t n; hg_geom g;
    if (!get_i32(env, a[1], &kind)) return NULL;
    double *m = (double *)get_typed(env, a[2], napi_float64_array, &n, "matrix"); if (!m) return NULL;
    if (n < (size_t)(kind == HG_AFFINE ? 6 : 8)) return throw_str(env, "hgwarp: matrix too short");
    if (!get_geom(env, a + 3, &g)) return NULL;
    const size_t px = (g.obj_w > 0 && g.obj_h > 0) ? (size_t)g.obj_w * g.obj_h : 0;
    void *out; napi_value r = make_pixels(env, px * 4, a[7], reuse, &out); if (!r) return NULL;
    if (px) HG_CALL(h->ctx, "hg_warp_inverse_geometric", hg_warp_inverse_geometric(h->ctx, kind, m, g, (uint8_t *)out));
    return r;
}

static napi_value fn_piecewise_set_mesh(napi_env env, napi_callback_info info)
{
    napi_value a[5];
    if (!get_args(env, info, 5, a)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    size_t np, nt; int msx, msy;
    float *src = (float *)get_typed(env, a[1], napi_float32_array, &np, "srcPoints"); if (!src) return NULL;
    uint32_t *tris = (uint32_t *)get_typed(env, a[2], napi_uint32_array, &nt, "triangles"); if (!tris) return NULL;
    if (!get_i32(env, a[3], &msx) || !get_i32(env, a[4], &msy)) return NULL;
    HG_CALL(h->ctx, "hg_piecewise_set_mesh", hg_piecewise_set_mesh(h->ctx, src, (int)(np / 2), tris, (int)(nt / 3), msx, msy));
    h->n_pts = np / 2; h->n_tris = nt / 3;
    return NULL;
}

static napi_value fn_piecewise_prepare(napi_env env, napi_callback_info info)
{
    napi_value a[6];
    if (!get_args(env, info, 6, a)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    size_t n; hg_geom g;
    float *dst = (float *)get_typed(env, a[1], napi_float32_array, &n, "dstPoints"); if (!dst) return NULL;
    if (!get_geom(env, a + 2, &g)) return NULL;
    if (h->n_pts == 0 || n < 2 * h->n_pts) return throw_str(env, "hgwarp: dstPoints must hold one x,y pair per mesh point (piecewiseSetMesh first)");
    HG_CALL(h->ctx, "hg_piecewise_prepare", hg_piecewise_prepare(h->ctx, dst, g));
    h->obj_w = g.obj_w; h->obj_h = g.obj_h;
    return NULL;
}

static napi_value fn_warp_inverse_piecewise(napi_env env, napi_callback_info info)
{
    napi_value a[2]; int reuse = 0;
    if (!get_args_opt(env, info, 1, a, &reuse)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    const size_t px = (h->obj_w > 0 && h->obj_h > 0) ? (size_t)h->obj_w * h->obj_h : 0;
    void *out; napi_value r = make_pixels(env, px * 4, a[1], reuse, &out); if (!r) return NULL;
    if (px) HG_CALL(h->ctx, "hg_warp_inverse_piecewise", hg_warp_inverse_piecewise(h->ctx, (uint8_t *)out));
    return r;
}

static napi_value fn_warp_forward_geometric(napi_env env, napi_callback_info info)
{
    napi_value a[7];
    if (!get_args(env, info, 7, a)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    int kind; size_t n; hg_geom g;
    if (!get_i32(env, a[1], &kind)) return NULL;
    double *m = (double *)get_typed(env, a[2], napi_float64_array, &n, "matrix"); if (!m) return NULL;
    if (n < (size_t)(kind == HG_AFFINE ? 6 : 8)) return throw_str(env, "hgwarp: matrix too short");
    if (!get_geom(env, a + 3, &g)) return NULL;
    const size_t px = (g.obj_w > 0 && g.obj_h > 0) ? (size_t)g.obj_w * g.obj_h : 0;
    void *out; napi_value r = make_typed(env, napi_uint8_clamped_array, px * 4, 1, &out); if (!r) return NULL;
    if (px) HG_CALL(h->ctx, "hg_warp_forward_geometric", hg_warp_forward_geometric(h->ctx, kind, m, g, (uint8_t *)out));
    return r;
}

static napi_value fn_warp_forward_piecewise(napi_env env, napi_callback_info info)
{
    napi_value a[8];
    if (!get_args(env, info, 8, a)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    size_t n; int mx, my; hg_geom g;
    float *dst = (float *)get_typed(env, a[1], napi_float32_array, &n, "dstPoints"); if (!dst) return NULL;
    if (!get_i32(env, a[2], &mx) || !get_i32(env, a[3], &my)) return NULL;
    if (!get_geom(env, a + 4, &g)) return NULL;
    if (h->n_pts == 0 || n < 2 * h->n_pts) return throw_str(env, "hgwarp: dstPoints must hold one x,y pair per mesh point (piecewiseSetMesh first)");
    const size_t px = (g.obj_w > 0 && g.obj_h > 0) ? (size_t)g.obj_w * g.obj_h : 0;
    void *out; napi_value r = make_typed(env, napi_uint8_clamped_array, px * 4, 1, &out); if (!r) return NULL;
    if (px) HG_CALL(h->ctx, "hg_warp_forward_piecewise", hg_warp_forward_piecewise(h->ctx, dst, mx, my, g, (uint8_t *)out));
    return r;
}

/* parity taps */
static napi_value fn_get_tri_map(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    bool fused = false;
    napi_get_value_bool(env, a[1], &fused);
    const size_t n = (h->obj_w > 0 && h->obj_h > 0) ? (size_t)h->obj_w * h->obj_h : 0;
    void *out; napi_value r = make_typed(env, napi_int16_array, n, 2, &out); if (!r) return NULL;
    if (n) HG_CALL(h->ctx, "hg_get_tri_map", fused ? hg_get_tri_map_fused(h->ctx, (int16_t *)out, n) : hg_get_tri_map(h->ctx, (int16_t *)out, n));
    return r;
}

static napi_value fn_get_matrices(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    int T; if (!get_i32(env, a[1], &T)) return NULL;
    T = (int)h->n_tris;                                      /* the C ABI fills n_triangles x 6 floats: size by the mesh, not by the caller */
    void *fwd, *inv;
    napi_value rf = make_typed(env, napi_float32_array, (size_t)T * 6, 4, &fwd); if (!rf) return NULL;
    napi_value ri = make_typed(env, napi_float32_array, (size_t)T * 6, 4, &inv); if (!ri) return NULL;
    if (T) HG_CALL(h->ctx, "hg_get_matrices", hg_get_matrices(h->ctx, (float *)fwd, (float *)inv));
    napi_value obj;
    NAPI_OK(napi_create_object(env, &obj));
    NAPI_OK(napi_set_named_property(env, obj, "forward", rf));
    NAPI_OK(napi_set_named_property(env, obj, "inverse", ri));
    return obj;
}

/* The caller loop `setDestinyPoints(dst_f); warp()` for F frames in one device pass: dst = F x 2N float32,
 * geoms = Int32Array F x 4 (xOff, yOff, objW, objH); returns an Array of F Uint8ClampedArray. */
static napi_value fn_warp_inverse_piecewise_batch(napi_env env, napi_callback_info info)
{
    napi_value a[3];
    if (!get_args(env, info, 3, a)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    size_t nd, ng;
    float *dst = (float *)get_typed(env, a[1], napi_float32_array, &nd, "dstPoints"); if (!dst) return NULL;
    int32_t *gv = (int32_t *)get_typed(env, a[2], napi_int32_array, &ng, "geoms"); if (!gv) return NULL;
    const int F = (int)(ng / 4);
    if (F <= 0) return throw_str(env, "hgwarp: geoms must hold 4 integers per frame");
    if (h->n_pts == 0 || nd < (size_t)F * 2 * h->n_pts) return throw_str(env, "hgwarp: dstPoints must hold frames x mesh points x,y pairs (piecewiseSetMesh first)");
    size_t *offs = (size_t *)malloc(sizeof(size_t) * F);
    size_t total = 0;
    hg_pack_offsets((const hg_geom *)gv, F, offs, &total);
    int rc = HG_OK;
    if (total > h->d_batch_cap) {
        if (h->d_batch) hg_device_free(h->ctx, h->d_batch);
        h->d_batch = NULL; h->d_batch_cap = 0;
        rc = hg_device_alloc(h->ctx, total, &h->d_batch);
        if (rc == HG_OK) h->d_batch_cap = total;
    }
    void *d_out = h->d_batch;
    if (rc == HG_OK) rc = hg_warp_inverse_piecewise_batch_device(h->ctx, dst, (const hg_geom *)gv, offs, F, d_out);
    if (rc == HG_OK) rc = hg_sync(h->ctx);
    napi_value arr = NULL;
    if (rc == HG_OK && napi_create_array_with_length(env, F, &arr) == napi_ok) {
        for (int f = 0; f < F && rc == HG_OK; f++) {
            const hg_geom *g = (const hg_geom *)gv + f;
            const size_t px = (g->obj_w > 0 && g->obj_h > 0) ? (size_t)g->obj_w * g->obj_h : 0;
            void *out; napi_value ta = make_typed(env, napi_uint8_clamped_array, px * 4, 1, &out);
            if (!ta) { rc = HG_ERR_NOMEM; break; }
            if (px) rc = hg_copy_to_host(h->ctx, out, (const uint8_t *)d_out + offs[f], px * 4);
            napi_set_element(env, arr, f, ta);
        }
    }
    free(offs);
    if (rc != HG_OK) return throw_hg(env, h->ctx, "warpInversePiecewiseBatch", rc);
    return arr;
}

/* ---------------------------------------------------------------- module */
static napi_value init(napi_env env, napi_value exports)
{
    static const struct { const char *name; napi_callback fn; } fns[] = {
        { "create", fn_create }, { "destroy", fn_destroy }, { "deviceCount", fn_device_count },
        { "solveAffine", fn_solve_affine }, { "invertAffine", fn_invert_affine }, { "solveProjective", fn_solve_projective },
        { "transformLimits", fn_transform_limits }, { "minmaxXY", fn_minmax_xy }, { "triangulate", fn_triangulate },
        { "setImage", fn_set_image }, { "warpInverseGeometric", fn_warp_inverse_geometric },
        { "piecewiseSetMesh", fn_piecewise_set_mesh }, { "piecewisePrepare", fn_piecewise_prepare },
        { "warpInversePiecewise", fn_warp_inverse_piecewise }, { "getTriMap", fn_get_tri_map }, { "getMatrices", fn_get_matrices },
        { "warpInversePiecewiseBatch", fn_warp_inverse_piecewise_batch },
        { "warpForwardGeometric", fn_warp_forward_geometric }, { "warpForwardPiecewise", fn_warp_forward_piecewise },
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
