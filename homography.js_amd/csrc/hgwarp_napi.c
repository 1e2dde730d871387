/*
 * hgwarp_napi.c -- thin N-API addon over the C ABI (include/hgwarp.h) for the drop-in JS class js/Homography.mjs.
 *
 * It only unwraps TypedArrays, calls libhgwarp.so on the main JS thread (the reference's warp() is synchronous too)
 * and creates the result arrays; failures are thrown as bare strings like the reference does (`throw("...")`).
 * Built with plain gcc against /usr/include/node (Makefile target lib/hgwarp.node); no node-gyp, no C++.
 */
#define _GNU_SOURCE
#include <node_api.h>
#include <sys/mman.h>
#include <stdint.h>
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
/* d_imgs: device buffer of the per-frame sources of setImages() (kept, grown as needed) */
/* slab: page-locked memory behind ONE external ArrayBuffer whose views are the frames of a batch (see "batches" below) */
typedef struct { void *ptr; size_t cap; napi_ref ab; } slab_t;
typedef struct { hg_ctx *ctx; int obj_w, obj_h; void *d_batch; size_t d_batch_cap; size_t n_pts, n_tris; void *d_imgs; size_t d_imgs_cap; slab_t slab[2]; } handle_t;

static void release_ctx(handle_t *h)
{
    if (h->ctx) {
        if (h->d_batch) hg_device_free(h->ctx, h->d_batch);
        if (h->d_imgs) hg_device_free(h->ctx, h->d_imgs);      /* (waits for the stream; the context that aliases it goes next) */
        hg_destroy(h->ctx);
    }
    h->ctx = NULL; h->d_batch = NULL; h->d_batch_cap = 0; h->d_imgs = NULL; h->d_imgs_cap = 0;
}

static void slab_drop(napi_env env, slab_t *sl, int detach);

static void handle_finalize(napi_env env, void *data, void *hint)
{
    (void)hint;
    handle_t *h = (handle_t *)data;
    if (h) { for (int k = 0; k < 2; k++) slab_drop(env, &h->slab[k], 0); release_ctx(h); free(h); }
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
    /* NaN, +-Infinity and anything outside int32 would be undefined behaviour in the conversion (degenerate matrices give
     * such window sizes; the reference then dies with a RangeError on its allocation) */
    if (!(d >= -2147483648.0 && d <= 2147483647.0)) { throw_str(env, "hgwarp: number is not finite or does not fit an int32 (Invalid typed array length)"); return 0; }
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

/* ---------------------------------------------------------------- pinned frame pool
 * A fresh 34 MB Uint8ClampedArray per 4K frame is what the reference's contract asks for (:991, :1040), and in V8 that is a
 * calloc -> mmap -> ~8 400 first-touch page faults (8-9 ms, far more than the 1.3 ms of PCIe traffic).  Frames of 1 MiB and
 * more therefore come from a pool of page-locked buffers (hg_host_alloc) handed to JavaScript as EXTERNAL ArrayBuffers: the
 * GPU DMAs straight into them and a buffer returns to the pool when its ArrayBuffer is collected.  Node 12 runs N-API
 * finalizers from the event loop only, so a loop that never yields gets no buffer back: the pool then grows up to
 * `pin_limit` bytes (default 2 GiB, setPinnedLimit()) and after that warp() falls back to plain V8 arrays (slower, still
 * correct).  To get buffers back inside such a loop anyway, every pooled frame also carries a WEAK reference to its ArrayBuffer:
 * V8 clears it during the collection itself, so pin_acquire() can see synchronously that a frame is dead and reuse its
 * buffer long before the deferred finalizer runs (and the JS class asks for a collection when poolPressure() says the pool is
 * starving: see there).  release(typedArray) hands a frame back at once (its ArrayBuffer is detached). */
typedef struct { void *ptr; size_t cap; int in_use; int zombie; unsigned gen; napi_ref weak; } pin_t;
typedef struct { int idx; unsigned gen; } pin_ref_t;
#define PIN_MAX 512
#define PIN_GC_EVERY 32
static const size_t PIN_MIN_FRAME = (size_t)1 << 20;
/* The pool is PER ENVIRONMENT (napi_set_instance_data): its napi_refs belong to one env and its buffers are handed out on that
 * env's thread only, so a second env (worker_threads, or the addon loaded into two contexts) gets a pool of its own instead of
 * racing on -- and dereferencing references of -- somebody else's. */
typedef struct {
    pin_t pins[PIN_MAX];
    int npins;
    size_t pin_bytes, pin_limit;
    int since_gc;                                            /* pooled frames handed out since the last requested collection */
    int pool_malloc;                                         /* _poolTestFrames(): plain malloc instead of pinned memory (no GPU needed) */
    /* May a buffer whose ArrayBuffer is dead (weak reference cleared) or detached (release()) back a NEW external ArrayBuffer
     * before the old one's finalizer has run?  Under Node 12 (V8 7.x) yes, and that is what lets a synchronous loop get frames
     * back at all (finalizers only run from the event loop there).  From V8 8 on (Node >= 14) two ArrayBuffers over one backing
     * pointer abort the process (nodejs/node#32463): there a buffer is recycled only once its finalizer HAS run. */
    int early_reuse;
    double stat_hit, stat_new, stat_fallback, stat_reaped, stat_finalized, stat_forced_gc;   /* poolStats() */
} pool_t;

static pool_t *pool_of(napi_env env)
{
    void *p = NULL;
    if (napi_get_instance_data(env, &p) != napi_ok || !p) { throw_str(env, "hgwarp: addon instance data is missing"); return NULL; }
    return (pool_t *)p;
}

static void pin_drop(pool_t *P, int i)
{
    if (P->pins[i].ptr) { if (P->pool_malloc) free(P->pins[i].ptr); else hg_host_free(P->pins[i].ptr); P->pin_bytes -= P->pins[i].cap; }
    P->pins[i].ptr = NULL; P->pins[i].cap = 0; P->pins[i].in_use = 0; P->pins[i].zombie = 0;
}

/* frames whose ArrayBuffer has been collected (weak reference cleared) although their finalizer has not run yet */
static void pin_reap(napi_env env, pool_t *P)
{
    for (int i = 0; i < P->npins; i++) {
        if (!P->pins[i].ptr || !P->pins[i].in_use || !P->pins[i].weak) continue;
        napi_value v = NULL;
        if (napi_get_reference_value(env, P->pins[i].weak, &v) == napi_ok && v == NULL) {
            napi_delete_reference(env, P->pins[i].weak);
            P->pins[i].weak = NULL;
            P->stat_reaped++;
            int64_t adj; napi_adjust_external_memory(env, -(int64_t)P->pins[i].cap, &adj);
            if (P->early_reuse) { P->pins[i].in_use = 0; P->pins[i].gen++; }   /* the late finalizer will not match */
            else P->pins[i].zombie = 1;                                        /* V8 >= 8: stays out of circulation until pin_finalize */
        }
    }
}

static int pin_acquire(napi_env env, pool_t *P, size_t bytes)
{
    if (bytes < PIN_MIN_FRAME || P->pin_limit == 0) return -1;
    int best = -1, empty = -1;
    for (int pass = 0; pass < 2 && best < 0; pass++) {
        if (pass == 1) pin_reap(env, P);
        for (int i = 0; i < P->npins; i++) {
            if (!P->pins[i].ptr) continue;
            if (!P->pins[i].in_use && P->pins[i].cap >= bytes && P->pins[i].cap <= bytes + bytes / 2 && (best < 0 || P->pins[i].cap < P->pins[best].cap)) best = i;
        }
    }
    P->since_gc++;
    if (best >= 0) { P->pins[best].in_use = 1; P->stat_hit++; return best; }
    for (int i = 0; i < P->npins; i++) if (!P->pins[i].ptr) { empty = i; break; }
    /* make room: idle buffers of the wrong size go first */
    for (int i = 0; i < P->npins && P->pin_bytes + bytes > P->pin_limit; i++) if (P->pins[i].ptr && !P->pins[i].in_use) { pin_drop(P, i); if (empty < 0) empty = i; }
    if (P->pin_bytes + bytes > P->pin_limit) { P->stat_fallback++; return -1; }
    if (empty < 0) { if (P->npins == PIN_MAX) { P->stat_fallback++; return -1; } empty = P->npins++; }
    void *p = NULL;
    if (P->pool_malloc) p = malloc(bytes);
    else if (hg_host_alloc(bytes, &p) != HG_OK) p = NULL;
    if (!p) { P->stat_fallback++; return -1; }
    P->stat_new++;
    P->pins[empty].ptr = p; P->pins[empty].cap = bytes; P->pins[empty].in_use = 1; P->pins[empty].gen++;
    P->pin_bytes += bytes;
    return empty;
}

static void pin_finalize(napi_env env, void *data, void *hint)
{
    (void)data;
    pin_ref_t *r = (pin_ref_t *)hint;
    pool_t *P = NULL;
    if (napi_get_instance_data(env, (void **)&P) != napi_ok || !P) { free(r); return; }      /* (the env is going away: pool_free owns the buffers) */
    if (r && r->idx >= 0 && r->idx < P->npins && P->pins[r->idx].gen == r->gen && P->pins[r->idx].in_use) {
        if (P->pins[r->idx].weak) { napi_delete_reference(env, P->pins[r->idx].weak); P->pins[r->idx].weak = NULL; }
        P->stat_finalized++;
        P->pins[r->idx].in_use = 0;
        if (P->pins[r->idx].zombie) P->pins[r->idx].zombie = 0;               /* (its external-memory accounting went when it was reaped / released) */
        else { int64_t adj; napi_adjust_external_memory(env, -(int64_t)P->pins[r->idx].cap, &adj); }
    }
    free(r);
}

/* a Uint8ClampedArray of `bytes` bytes over a pooled pinned buffer, or NULL (pool exhausted / small frame) */
static napi_value make_pinned(napi_env env, size_t bytes, void **data)
{
    pool_t *P = pool_of(env);
    if (!P) return NULL;
    const int i = pin_acquire(env, P, bytes);
    if (i < 0) return NULL;
    pin_ref_t *r = (pin_ref_t *)malloc(sizeof *r);
    napi_value ab, ta;
    if (!r) { P->pins[i].in_use = 0; return NULL; }
    r->idx = i; r->gen = P->pins[i].gen;
    if (napi_create_external_arraybuffer(env, P->pins[i].ptr, bytes, pin_finalize, r, &ab) != napi_ok) { P->pins[i].in_use = 0; free(r); return NULL; }
    int64_t adj; napi_adjust_external_memory(env, (int64_t)P->pins[i].cap, &adj);     /* V8 learns about the memory pressure */
    P->pins[i].weak = NULL;
    if (napi_create_reference(env, ab, 0, &P->pins[i].weak) != napi_ok) P->pins[i].weak = NULL;
    if (napi_create_typedarray(env, napi_uint8_clamped_array, bytes, ab, 0, &ta) != napi_ok) return NULL;
    *data = P->pins[i].ptr;
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
    napi_value pinned = make_pinned(env, bytes, data);        /* fresh frame, pooled page-locked memory (see above) */
    if (pinned) return pinned;
    napi_value ta = make_typed(env, napi_uint8_clamped_array, bytes, 1, data);
    /* Pool exhausted (or switched off): a plain V8 array.  Its memory is a fresh anonymous mapping nobody has touched yet;
     * asking for transparent huge pages BEFORE the first write turns ~8 400 4-KiB page faults of a 4K frame into ~17 (where
     * the kernel runs THP in `madvise` mode, as on the MI355X hosts).  Advice only: failure changes nothing. */
    if (ta && bytes >= ((size_t)4 << 20) && *data) {
        const uintptr_t lo = ((uintptr_t)*data + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)*data + bytes) & ~(uintptr_t)4095;
        if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
    }
    return ta;
}

/* release(typedArray): a frame produced by warp() goes back to the pool now; its ArrayBuffer is detached (length 0). */
static napi_value fn_release(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    bool is = false;
    napi_typedarray_type t; size_t n = 0, off = 0; void *p = NULL; napi_value ab;
    napi_value res; napi_get_boolean(env, false, &res);
    pool_t *P = pool_of(env); if (!P) return NULL;
    if (napi_is_typedarray(env, a[0], &is) != napi_ok || !is || napi_get_typedarray_info(env, a[0], &t, &n, &p, &ab, &off) != napi_ok || !p) return res;
    for (int i = 0; i < P->npins; i++)
        if (P->pins[i].ptr == (char *)p - off && P->pins[i].in_use) {
            if (napi_detach_arraybuffer(env, ab) != napi_ok) return res;
            if (P->pins[i].weak) { napi_delete_reference(env, P->pins[i].weak); P->pins[i].weak = NULL; }
            if (P->early_reuse) { P->pins[i].in_use = 0; P->pins[i].gen++; }   /* the pending finalizer of this ArrayBuffer no longer matches */
            else P->pins[i].zombie = 1;                                        /* V8 >= 8: recycled once the detached buffer's finalizer has run */
            int64_t adj; napi_adjust_external_memory(env, -(int64_t)P->pins[i].cap, &adj);
            napi_get_boolean(env, true, &res);
            break;
        }
    return res;
}

/* _poolTestFrames(n, bytes): n frames of `bytes` bytes through the very allocation path warp() / warpBatch() use, without any
 * GPU work (tests of the pool's life cycle on machines without a GPU; the pool is switched to plain malloc memory). */
static napi_value fn_pool_test_frames(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    int n; double bytes;
    if (!get_i32(env, a[0], &n) || napi_get_value_double(env, a[1], &bytes) != napi_ok || n < 0 || !(bytes >= 0)) return throw_str(env, "hgwarp: _poolTestFrames(n, bytes)");
    pool_t *P = pool_of(env); if (!P) return NULL;
    if (!P->pool_malloc) { for (int i = 0; i < P->npins; i++) if (P->pins[i].ptr && !P->pins[i].in_use) pin_drop(P, i); P->pool_malloc = 1; }
    napi_value arr;
    NAPI_OK(napi_create_array_with_length(env, n, &arr));
    for (int f = 0; f < n; f++) {
        void *out; napi_value ta = make_pixels(env, (size_t)bytes, NULL, 0, &out);
        if (!ta) return NULL;
        if (bytes > 0) ((uint8_t *)out)[0] = (uint8_t)f;
        napi_set_element(env, arr, f, ta);
    }
    return arr;
}

/* pinnedBuffer(bytes): a zeroed Uint8ClampedArray in pooled page-locked memory (a plain V8 array when the pool is off, exhausted, or the
 * buffer is small): for SOURCE images the caller fills itself (decoded video frames ...) -- uploads out of page-locked memory are true
 * asynchronous DMA at the full PCIe rate, uploads out of V8's pageable memory are staged by the runtime (~20 % slower and host-blocking). */
static napi_value fn_pinned_buffer(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    double bytes;
    if (napi_get_value_double(env, a[0], &bytes) != napi_ok || !(bytes >= 0) || bytes > 4.0e9) return throw_str(env, "hgwarp: pinnedBuffer(bytes)");
    void *out = NULL;
    napi_value ta = make_pixels(env, (size_t)bytes, NULL, 0, &out);
    if (ta && out && bytes > 0) memset(out, 0, (size_t)bytes);
    return ta;
}

/* poolPressure(bytes, n): should the caller run a collection before asking for n frames of `bytes` bytes?  True when the pool
 * cannot serve them from free (or already collected) buffers AND at least `g_next_gc_live` pooled frames are alive or the
 * cap would be exceeded.  A loop that allocates little JavaScript (warpBatch: 8 frames per call) can go a long time without
 * a collection although dozens of 34 MB frames are garbage; V8's own external-memory heuristics start incremental cycles
 * that treat everything allocated meanwhile as live.  The JS class then calls a real collector (js/Homography.mjs) and
 * poolCollected(), which reaps and restarts the count: the next collection is due after another PIN_GC_EVERY pooled frames
 * (a caller that really keeps all its frames alive pays one useless collection per PIN_GC_EVERY frames), or when the cap is hit. */

static napi_value fn_pool_pressure(napi_env env, napi_callback_info info)
{
    napi_value a[2], res;
    if (!get_args(env, info, 2, a)) return NULL;
    double bytes; int n;
    napi_get_boolean(env, false, &res);
    pool_t *P = pool_of(env); if (!P) return NULL;
    if (napi_get_value_double(env, a[0], &bytes) != napi_ok || !get_i32(env, a[1], &n)) return res;
    if (!(bytes >= (double)PIN_MIN_FRAME) || P->pin_limit == 0 || n <= 0) return res;
    pin_reap(env, P);
    int fit = 0, live = 0;
    for (int i = 0; i < P->npins; i++) {
        if (!P->pins[i].ptr) continue;
        if (P->pins[i].in_use) live++;
        else if (P->pins[i].cap >= (size_t)bytes && P->pins[i].cap <= (size_t)bytes + (size_t)bytes / 2) fit++;
    }
    if (fit >= n) return res;
    (void)live;
    if (P->since_gc >= PIN_GC_EVERY || (double)P->pin_bytes + (n - fit) * bytes > (double)P->pin_limit) napi_get_boolean(env, true, &res);
    return res;
}

static napi_value fn_pool_collected(napi_env env, napi_callback_info info)
{
    (void)info;
    pool_t *P = pool_of(env); if (!P) return NULL;
    const double before = P->stat_reaped;
    pin_reap(env, P);
    P->since_gc = 0;
    P->stat_forced_gc++;
    napi_value v; NAPI_OK(napi_create_double(env, P->stat_reaped - before, &v));
    return v;
}

/* poolStats(): counters of the pinned frame pool since the addon was loaded */
static napi_value fn_pool_stats(napi_env env, napi_callback_info info)
{
    (void)info;
    napi_value o, v;
    pool_t *P = pool_of(env); if (!P) return NULL;
    NAPI_OK(napi_create_object(env, &o));
    int in_use = 0, total = 0;
    for (int i = 0; i < P->npins; i++) if (P->pins[i].ptr) { total++; in_use += P->pins[i].in_use; }
    const struct { const char *k; double x; } kv[] = { {"pinnedBytes", (double)P->pin_bytes}, {"buffers", total}, {"inUse", in_use}, {"reused", P->stat_hit},
        {"allocated", P->stat_new}, {"fallbackToV8", P->stat_fallback}, {"reapedByWeakRef", P->stat_reaped}, {"forcedCollections", P->stat_forced_gc}, {"finalized", P->stat_finalized}, {"earlyReuse", P->early_reuse} };
    for (size_t i = 0; i < sizeof kv / sizeof kv[0]; i++) { NAPI_OK(napi_create_double(env, kv[i].x, &v)); NAPI_OK(napi_set_named_property(env, o, kv[i].k, v)); }
    return o;
}

/* setPinnedLimit(bytes): cap of the pinned frame pool (0 disables it: every frame is a plain V8 array); returns bytes in use */
static napi_value fn_set_pinned_limit(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    double d;
    if (napi_get_value_double(env, a[0], &d) != napi_ok || !(d >= 0)) return throw_str(env, "hgwarp: setPinnedLimit needs a non-negative number of bytes");
    pool_t *P = pool_of(env); if (!P) return NULL;
    P->pin_limit = d > 1e15 ? (size_t)1e15 : (size_t)d;
    for (int i = 0; i < P->npins; i++) if (P->pins[i].ptr && !P->pins[i].in_use && P->pin_bytes > P->pin_limit) pin_drop(P, i);
    napi_value v;
    NAPI_OK(napi_create_double(env, (double)P->pin_bytes, &v));
    return v;
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
    if (napi_get_value_external(env, a[0], &p) == napi_ok && p) {
        handle_t *h = (handle_t *)p;
        if (h->ctx) (void)hg_sync(h->ctx);
        for (int k = 0; k < 2; k++) slab_drop(env, &h->slab[k], 0);      /* (frames the caller still holds keep their memory: the ArrayBuffer owns it) */
        release_ctx(h);
    }
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

/* setImages(ctx, [Uint8ClampedArray, ...], w, h): one source per frame of the next batch (`warp(image_f)` per frame in the
 * reference: setImage :290 on every call).  The images are copied into one device buffer; frame f reads image f % n. */
static napi_value fn_set_images(napi_env env, napi_callback_info info)
{
    napi_value a[4];
    if (!get_args(env, info, 4, a)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    bool is_arr = false; uint32_t n = 0; int w, hh;
    if (napi_is_array(env, a[1], &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, a[1], &n) != napi_ok || n == 0)
        return throw_str(env, "hgwarp: setImages needs a non-empty array of image data arrays");
    if (!get_i32(env, a[2], &w) || !get_i32(env, a[3], &hh)) return NULL;
    if (w <= 0 || hh <= 0) return throw_str(env, "hgwarp: bad image size");
    const size_t bytes = (size_t)w * (size_t)hh * 4, stride = (bytes + 255) & ~(size_t)255;
    if (stride * n > h->d_imgs_cap) {
        /* grow: new buffer first, the context's alias moves to it, only then the old one is freed (never a dangling alias) */
        void *q = NULL;
        HG_CALL(h->ctx, "hg_device_alloc", hg_device_alloc(h->ctx, stride * n, &q));
        int rc = hg_set_images_device(h->ctx, q, w, hh, (int)n, stride);
        if (rc != HG_OK) { hg_device_free(h->ctx, q); return throw_hg(env, h->ctx, "hg_set_images_device", rc); }
        if (h->d_imgs) hg_device_free(h->ctx, h->d_imgs);
        h->d_imgs = q; h->d_imgs_cap = stride * n;
    }
    for (uint32_t k = 0; k < n; k++) {
        napi_value el; size_t len;
        NAPI_OK(napi_get_element(env, a[1], k, &el));
        uint8_t *px = (uint8_t *)get_typed(env, el, napi_uint8_clamped_array, &len, "images[k]"); if (!px) return NULL;
        if (len < bytes) return throw_str(env, "hgwarp: an image is smaller than width*height*4");
        HG_CALL(h->ctx, "hg_copy_to_device", hg_copy_to_device(h->ctx, (uint8_t *)h->d_imgs + stride * k, px, bytes));
    }
    HG_CALL(h->ctx, "hg_set_images_device", hg_set_images_device(h->ctx, h->d_imgs, w, hh, (int)n, stride));
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
    void *out; napi_value r = make_pixels(env, px * 4, NULL, 0, &out); if (!r) return NULL;
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
    void *out; napi_value r = make_pixels(env, px * 4, NULL, 0, &out); if (!r) return NULL;
    if (px) HG_CALL(h->ctx, "hg_warp_forward_piecewise", hg_warp_forward_piecewise(h->ctx, dst, mx, my, g, (uint8_t *)out));
    return r;
}

/* ---------------------------------------------------------------- reference-state forms (stale matrices / stale map) */
/* solveAffineTriangles(src, dst, triangles) -> Float32Array(6 T): affineMatrixFromTriangles per triangle (:785-804), on the host */
static napi_value fn_solve_affine_triangles(napi_env env, napi_callback_info info)
{
    napi_value a[3];
    if (!get_args(env, info, 3, a)) return NULL;
    size_t ns, nd, nt;
    float *s = (float *)get_typed(env, a[0], napi_float32_array, &ns, "src"); if (!s) return NULL;
    float *d = (float *)get_typed(env, a[1], napi_float32_array, &nd, "dst"); if (!d) return NULL;
    uint32_t *t = (uint32_t *)get_typed(env, a[2], napi_uint32_array, &nt, "triangles"); if (!t) return NULL;
    const size_t np = (ns < nd ? ns : nd) / 2, T = nt / 3;
    void *out; napi_value r = make_typed(env, napi_float32_array, 6 * T, 4, &out); if (!r) return NULL;
    HG_CALL(NULL, "hg_solve_affine_triangles", hg_solve_affine_triangles(s, d, (int)np, t, (int)T, (float *)out));
    return r;
}

/* a map id without a matrix: the reference's loop throws a TypeError at that pixel (`matrix[0]` of undefined, :1383) */
static napi_value throw_state(napi_env env, hg_ctx *ctx, const char *what, int code)
{
    if (code == HG_ERR_RANGE) { napi_throw_type_error(env, NULL, "Cannot read property '0' of undefined"); return NULL; }
    return throw_hg(env, ctx, what, code);
}

/* warpInversePiecewiseState(ctx, fwdMats, dstPoints, triangles, minSrcX, minSrcY, xOff, yOff, objW, objH) */
static napi_value fn_warp_inverse_piecewise_state(napi_env env, napi_callback_info info)
{
    napi_value a[10];
    if (!get_args(env, info, 10, a)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    size_t nm, np, nt; int msx, msy; hg_geom g;
    float *m = (float *)get_typed(env, a[1], napi_float32_array, &nm, "matrices"); if (!m) return NULL;
    float *p = (float *)get_typed(env, a[2], napi_float32_array, &np, "dstPoints"); if (!p) return NULL;
    uint32_t *t = (uint32_t *)get_typed(env, a[3], napi_uint32_array, &nt, "triangles"); if (!t) return NULL;
    if (!get_i32(env, a[4], &msx) || !get_i32(env, a[5], &msy)) return NULL;
    if (!get_geom(env, a + 6, &g)) return NULL;
    const size_t px = (g.obj_w > 0 && g.obj_h > 0) ? (size_t)g.obj_w * g.obj_h : 0;
    void *out; napi_value r = make_pixels(env, px * 4, NULL, 0, &out); if (!r) return NULL;
    hg_tri_map_def def = { p, (int)(np / 2), t, (int)(nt / 3), g.obj_w, g.obj_h, g.y_off };
    if (px) { int rc = hg_warp_inverse_piecewise_state(h->ctx, m, (int)(nm / 6), &def, msx, msy, g, (uint8_t *)out);
              if (rc != HG_OK) return throw_state(env, h->ctx, "hg_warp_inverse_piecewise_state", rc); }
    return r;
}

/* warpForwardPiecewiseState(ctx, fwdMats, mapPoints, mapTriangles, mapWidth, mapHeight, mapYOff, minSrcX, minSrcY, maxSrcX, maxSrcY, xOff, yOff, objW, objH) */
static napi_value fn_warp_forward_piecewise_state(napi_env env, napi_callback_info info)
{
    napi_value a[15];
    if (!get_args(env, info, 15, a)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    size_t nm, np, nt; int mw, mh, myo, msx, msy, mxx, mxy; hg_geom g;
    float *m = (float *)get_typed(env, a[1], napi_float32_array, &nm, "matrices"); if (!m) return NULL;
    float *p = (float *)get_typed(env, a[2], napi_float32_array, &np, "mapPoints"); if (!p) return NULL;
    uint32_t *t = (uint32_t *)get_typed(env, a[3], napi_uint32_array, &nt, "mapTriangles"); if (!t) return NULL;
    if (!get_i32(env, a[4], &mw) || !get_i32(env, a[5], &mh) || !get_i32(env, a[6], &myo)) return NULL;
    if (!get_i32(env, a[7], &msx) || !get_i32(env, a[8], &msy) || !get_i32(env, a[9], &mxx) || !get_i32(env, a[10], &mxy)) return NULL;
    if (!get_geom(env, a + 11, &g)) return NULL;
    const size_t px = (g.obj_w > 0 && g.obj_h > 0) ? (size_t)g.obj_w * g.obj_h : 0;
    void *out; napi_value r = make_pixels(env, px * 4, NULL, 0, &out); if (!r) return NULL;
    hg_tri_map_def def = { p, (int)(np / 2), t, (int)(nt / 3), mw, mh, myo };
    uint8_t none[4];                                          /* a blank window still gets the held map's ids checked (the reference's loop runs and may throw) */
    if (!px) { g.x_off = g.y_off = g.obj_w = g.obj_h = 0; }
    int rc = hg_warp_forward_piecewise_state(h->ctx, m, (int)(nm / 6), &def, msx, msy, mxx, mxy, g, px ? (uint8_t *)out : none);
    if (rc != HG_OK) return throw_state(env, h->ctx, "hg_warp_forward_piecewise_state", rc);
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

/* redoneFrames(ctx): frames the fused kernels flagged and hg_sync redid through the map so far (hg_redone_frames; tests) */
static napi_value fn_redone_frames(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    napi_value out;
    NAPI_OK(napi_create_double(env, (double)hg_redone_frames(h->ctx), &out));
    return out;
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

/* ---------------------------------------------------------------- batches: the caller loop `setDestinyPoints(dst_f); warp()` as device passes
 * Frames come back as views of ONE page-locked slab owned by the context handle (two of them: slot 0 for the frames warp() sends down
 * the inverse loops, slot 1 for the forward loops): one external ArrayBuffer per slab, created once and reused by the NEXT batch of the
 * same kind on this instance -- frames of a batch are valid until then (or releaseBatch()), like `reuseOutput` for warp().  Nothing
 * depends on the garbage collector: no per-frame ArrayBuffer, no finalizer has to run before memory comes back, no fall-back to V8
 * arrays.  `ownFrames` (optional argument) = the old behaviour: every frame its own pooled buffer with a GC-bound life time.
 * With per-frame sources ({images}: the video loop warp(image_f), README.md:121-137) the pass is pipelined frame by frame:
 * H2D(f + 1) on the context's copy stream runs while D2H(f) travels the other way on the warp stream. */
static void slab_finalize(napi_env env, void *data, void *hint)
{
    const size_t bytes = hint ? *(size_t *)hint : 0;          /* (the hint is the slab's size: what was added to V8's external-memory counter and to the pinned total) */
    free(hint);
    if (data) hg_host_free(data);
    pool_t *P = NULL;
    if (!bytes || napi_get_instance_data(env, (void **)&P) != napi_ok || !P) return;        /* (the env is going away) */
    P->pin_bytes -= bytes < P->pin_bytes ? bytes : P->pin_bytes;
    int64_t adj; napi_adjust_external_memory(env, -(int64_t)bytes, &adj);
}

static void slab_drop(napi_env env, slab_t *sl, int detach)
{
    if (sl->ab) {
        if (detach) { napi_value ab = NULL; if (napi_get_reference_value(env, sl->ab, &ab) == napi_ok && ab) (void)napi_detach_arraybuffer(env, ab); }
        napi_delete_reference(env, sl->ab);                   /* the ArrayBuffer's own finalizer frees the page-locked memory once no view is left */
    }
    sl->ab = NULL; sl->ptr = NULL; sl->cap = 0;
}

/* JS array of F Uint8ClampedArray views, frame f at byte offs[f] of slab `slot` (grown to `total` bytes if needed) */
/* The slab is page-locked memory like the pooled frames and counts against the same cap (setPinnedLimit): when it does not fit -- or the
 * cap is 0, or the allocation fails -- *fallback is set, NULL is returned WITHOUT a pending exception, and the batch gets own frames. */
static napi_value slab_frames(napi_env env, handle_t *h, int slot, const hg_geom *g, const size_t *offs, int F, size_t total, uint8_t **base, int *fallback)
{
    slab_t *sl = &h->slab[slot];
    *fallback = 0;
    pool_t *P = pool_of(env); if (!P) return NULL;
    if (sl->ab && (P->pin_limit == 0 || P->pin_bytes > P->pin_limit)) {      /* the cap was lowered since this slab was made: it goes (its views keep their memory until collected) */
        slab_drop(env, sl, 0);
        if (P->pin_limit == 0) { P->stat_fallback++; *fallback = 1; return NULL; }
    }
    if (total > sl->cap || !sl->ab) {
        slab_drop(env, sl, 0);
        const size_t want = total + total / 8 + 4096;
        for (int i = 0; i < P->npins && P->pin_bytes + want > P->pin_limit; i++) if (P->pins[i].ptr && !P->pins[i].in_use) pin_drop(P, i);   /* idle pooled frames make room */
        if (P->pin_limit == 0 || P->pin_bytes + want > P->pin_limit) { P->stat_fallback++; *fallback = 1; return NULL; }
        void *p = NULL; napi_value ab;
        size_t *hint = (size_t *)malloc(sizeof *hint);
        if (!hint || hg_host_alloc(want, &p) != HG_OK) { free(hint); P->stat_fallback++; *fallback = 1; return NULL; }
        *hint = want;
        if (napi_create_external_arraybuffer(env, p, want, slab_finalize, hint, &ab) != napi_ok) { hg_host_free(p); free(hint); return throw_str(env, "hgwarp: cannot wrap the frame slab"); }
        P->pin_bytes += want;                                 /* (slab_finalize takes both back) */
        int64_t adj; napi_adjust_external_memory(env, (int64_t)want, &adj);
        if (napi_create_reference(env, ab, 1, &sl->ab) != napi_ok) return throw_str(env, "hgwarp: cannot keep the frame slab");
        sl->ptr = p; sl->cap = want;
    }
    napi_value ab, arr;
    NAPI_OK(napi_get_reference_value(env, sl->ab, &ab));
    NAPI_OK(napi_create_array_with_length(env, F, &arr));
    for (int f = 0; f < F; f++) {
        const size_t px = (g[f].obj_w > 0 && g[f].obj_h > 0) ? (size_t)g[f].obj_w * g[f].obj_h : 0;
        napi_value ta;
        NAPI_OK(napi_create_typedarray(env, napi_uint8_clamped_array, px * 4, ab, px ? offs[f] : 0, &ta));
        NAPI_OK(napi_set_element(env, arr, f, ta));
    }
    *base = (uint8_t *)sl->ptr;
    return arr;
}

/* releaseBatch(ctx): the frames of the last batches become empty (their ArrayBuffers are detached), the slabs go */
static napi_value fn_release_batch(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    (void)hg_sync(h->ctx);
    for (int k = 0; k < 2; k++) slab_drop(env, &h->slab[k], 1);
    return NULL;
}

enum { JOB_PW_INV, JOB_GEO_INV, JOB_PW_FWD, JOB_GEO_FWD };
typedef struct { int type, kind; const float *dst, *from, *to; const double *m; int mx, my; size_t per; } batch_job;

/* frames [f0, f0 + n) of the job into d_out at offs[f0 ..] */
static int job_launch(handle_t *h, const batch_job *j, const hg_geom *g, const size_t *offs, int f0, int n, void *d_out)
{
    switch (j->type) {
    case JOB_PW_INV:  return hg_warp_inverse_piecewise_batch_device(h->ctx, j->dst + (size_t)f0 * 2 * h->n_pts, g + f0, offs + f0, n, d_out);
    case JOB_PW_FWD:  return hg_warp_forward_piecewise_batch_device(h->ctx, j->dst + (size_t)f0 * 2 * h->n_pts, j->mx, j->my, g + f0, offs + f0, n, d_out);
    case JOB_GEO_FWD: return hg_warp_forward_geometric_batch_device(h->ctx, j->kind, j->m + (size_t)f0 * 8, g + f0, offs + f0, n, d_out);
    default: {
        int rc = hg_geometric_set_frames_points(h->ctx, j->kind, j->from + (size_t)f0 * j->per, j->to + (size_t)f0 * j->per, g + f0, offs + f0, n);
        return rc == HG_OK ? hg_warp_inverse_geometric_frames_device(h->ctx, d_out) : rc;
    }
    }
}

/* opt[0] = ownFrames (boolean), opt[1] = images (Array of Uint8ClampedArray) or null / undefined, opt[2], opt[3] = their width, height */
static napi_value run_batch(napi_env env, handle_t *h, const batch_job *job, const int32_t *gv, int F, int slot, const napi_value *opt, size_t n_opt, const char *what)
{
    const hg_geom *g = (const hg_geom *)gv;
    bool own = false;
    if (n_opt > 0) { napi_valuetype vt; if (napi_typeof(env, opt[0], &vt) == napi_ok && vt == napi_boolean) napi_get_value_bool(env, opt[0], &own); }
    uint32_t n_img = 0;
    int iw = 0, ih = 0;
    if (n_opt > 1) {
        bool is_arr = false;
        if (napi_is_array(env, opt[1], &is_arr) == napi_ok && is_arr) {
            NAPI_OK(napi_get_array_length(env, opt[1], &n_img));
            if (n_img == 0 || n_opt < 4 || !get_i32(env, opt[2], &iw) || !get_i32(env, opt[3], &ih) || iw <= 0 || ih <= 0)
                return throw_str(env, "hgwarp: a batch with per-frame sources needs a non-empty image list and their width, height");
        }
    }
    size_t *offs = (size_t *)malloc(sizeof(size_t) * F);
    if (!offs) return throw_str(env, "hgwarp: out of memory (frame offsets)");
    size_t total = 0;
    hg_pack_offsets(g, F, offs, &total);
    int rc = HG_OK;
    if (total > h->d_batch_cap) {
        if (h->d_batch) hg_device_free(h->ctx, h->d_batch);
        h->d_batch = NULL; h->d_batch_cap = 0;
        rc = hg_device_alloc(h->ctx, total, &h->d_batch);
        if (rc == HG_OK) h->d_batch_cap = total;
    }
    if (rc != HG_OK) { free(offs); return throw_hg(env, h->ctx, what, rc); }
    /* where the frames go */
    napi_value arr = NULL;
    uint8_t *base = NULL;
    uint8_t **outs = NULL;
    if (!own) {
        int fallback = 0;
        arr = slab_frames(env, h, slot, g, offs, F, total, &base, &fallback);
        if (!arr && !fallback) { free(offs); return NULL; }
        if (!arr) own = true;                                 /* no room under the pinned cap (or no page-locked memory): own frames, as without the option */
    }
    if (own) {
        outs = (uint8_t **)calloc((size_t)F, sizeof *outs);
        if (!outs || napi_create_array_with_length(env, F, &arr) != napi_ok) { free(offs); free(outs); return throw_str(env, "hgwarp: out of memory (frames)"); }
        for (int f = 0; f < F; f++) {
            const size_t px = (g[f].obj_w > 0 && g[f].obj_h > 0) ? (size_t)g[f].obj_w * g[f].obj_h : 0;
            void *out; napi_value ta = make_pixels(env, px * 4, NULL, 0, &out);
            if (!ta) { free(offs); free(outs); return NULL; }
            outs[f] = (uint8_t *)out;
            napi_set_element(env, arr, f, ta);
        }
    }
    const uint8_t *d_out = (const uint8_t *)h->d_batch;
    if (n_img == 0) {
        /* one source for all frames: one pass; the copies queue behind the kernels on the warp stream (page-locked memory: asynchronous DMA) */
        rc = job_launch(h, job, g, offs, 0, F, h->d_batch);
        if (rc == HG_OK && !own) rc = hg_copy_to_host_async(h->ctx, base, d_out, total);       /* (the slab has the device layout: one copy) */
        for (int f = 0; f < F && rc == HG_OK && own; f++) {
            const size_t px = (g[f].obj_w > 0 && g[f].obj_h > 0) ? (size_t)g[f].obj_w * g[f].obj_h : 0;
            if (px) rc = hg_copy_to_host_async(h->ctx, outs[f], d_out + offs[f], px * 4);
        }
    } else {
        /* one source per frame (frame f reads images[f % n]): upload(f + 1) on the copy stream while frame f's pixels come down */
        const size_t bytes = (size_t)iw * (size_t)ih * 4, stride = (bytes + 255) & ~(size_t)255;
        const uint8_t **src = (const uint8_t **)calloc(n_img, sizeof *src);
        if (!src) rc = HG_ERR_NOMEM;
        for (uint32_t k = 0; k < n_img && rc == HG_OK; k++) {
            napi_value el; size_t len;
            if (napi_get_element(env, opt[1], k, &el) != napi_ok) { rc = HG_ERR_INVALID; break; }
            src[k] = (const uint8_t *)get_typed(env, el, napi_uint8_clamped_array, &len, "images[k]");
            if (!src[k]) { free(src); free(offs); free(outs); return NULL; }
            if (len < bytes) { free(src); free(offs); free(outs); return throw_str(env, "hgwarp: an image is smaller than width*height*4"); }
        }
        if (rc == HG_OK && stride * n_img > h->d_imgs_cap) {
            void *q = NULL;
            rc = hg_device_alloc(h->ctx, stride * n_img, &q);
            if (rc == HG_OK) rc = hg_set_images_device(h->ctx, q, iw, ih, 1, stride);      /* (the context's alias moves first: never dangling) */
            if (rc == HG_OK) { if (h->d_imgs) hg_device_free(h->ctx, h->d_imgs); h->d_imgs = q; h->d_imgs_cap = stride * n_img; }
            else if (q) hg_device_free(h->ctx, q);
        }
        /* Three streams: image f + 1 goes up (copy stream) and frame f - 1 comes down (download stream) while frame f is warped; the host
         * only ever waits for warps.  A frame comes down unsettled: if the settlement that precedes the next image binding (or the final
         * hg_sync) redid it -- hg_redone_frames moved --, it comes down again. */
        if (rc == HG_OK) rc = hg_upload_on_copy_stream(h->ctx, h->d_imgs, src[0], bytes);
        long redone = hg_redone_frames(h->ctx);
        for (int f = 0; f <= F && rc == HG_OK; f++) {
            if (f < F) rc = hg_set_images_device(h->ctx, (uint8_t *)h->d_imgs + stride * ((uint32_t)f % n_img), iw, ih, 1, stride);   /* (settles frame f - 1) */
            else rc = hg_sync(h->ctx);
            if (rc == HG_OK && f > 0) {
                const long now = hg_redone_frames(h->ctx);
                const size_t ppx = (g[f - 1].obj_w > 0 && g[f - 1].obj_h > 0) ? (size_t)g[f - 1].obj_w * g[f - 1].obj_h : 0;
                if (now != redone && ppx) rc = hg_download_behind_warps(h->ctx, own ? outs[f - 1] : base + offs[f - 1], d_out + offs[f - 1], ppx * 4);
                redone = now;
            } else if (rc == HG_OK) redone = hg_redone_frames(h->ctx);      /* f == 0: that binding settled runs queued BEFORE this batch; their redos are not frame 0's */
            if (f == F || rc != HG_OK) break;
            const size_t px = (g[f].obj_w > 0 && g[f].obj_h > 0) ? (size_t)g[f].obj_w * g[f].obj_h : 0;
            rc = hg_fence_copies(h->ctx);                                                   /* the warp stream waits for image f */
            if (rc == HG_OK) rc = job_launch(h, job, g, offs, f, 1, h->d_batch);
            if (rc == HG_OK && px) rc = hg_download_behind_warps(h->ctx, own ? outs[f] : base + offs[f], d_out + offs[f], px * 4);
            if (rc == HG_OK && (uint32_t)(f + 1) < n_img && f + 1 < F)                      /* ... and image f + 1 goes up meanwhile */
                rc = hg_upload_on_copy_stream(h->ctx, (uint8_t *)h->d_imgs + stride * (uint32_t)(f + 1), src[f + 1], bytes);
        }
        free(src);
        { const int rcd = hg_fence_downloads(h->ctx); if (rc == HG_OK) rc = rcd; }
    }
    const int rc2 = hg_sync(h->ctx);
    if (rc == HG_OK) rc = rc2;
    free(offs); free(outs);
    if (rc != HG_OK) return throw_hg(env, h->ctx, what, rc);
    return arr;
}

/* warpInversePiecewiseBatch(ctx, dstPoints F x 2N float32, geoms Int32Array F x 4 [, ownFrames, images, width, height]) */
static napi_value fn_warp_inverse_piecewise_batch(napi_env env, napi_callback_info info)
{
    napi_value a[7]; size_t argc = 7;
    if (napi_get_cb_info(env, info, &argc, a, NULL, NULL) != napi_ok || argc < 3) return throw_str(env, "hgwarp: wrong number of arguments");
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    size_t nd, ng; batch_job job = { JOB_PW_INV, 0, NULL, NULL, NULL, NULL, 0, 0, 0 };
    job.dst = (const float *)get_typed(env, a[1], napi_float32_array, &nd, "dstPoints"); if (!job.dst) return NULL;
    int32_t *gv = (int32_t *)get_typed(env, a[2], napi_int32_array, &ng, "geoms"); if (!gv) return NULL;
    const int F = (int)(ng / 4);
    if (F <= 0) return throw_str(env, "hgwarp: geoms must hold 4 integers per frame");
    if (h->n_pts == 0 || nd < (size_t)F * 2 * h->n_pts) return throw_str(env, "hgwarp: dstPoints must hold frames x mesh points x,y pairs (piecewiseSetMesh first)");
    return run_batch(env, h, &job, gv, F, 0, a + 3, argc - 3, "warpInversePiecewiseBatch");
}

/* warpInverseGeometricBatch(ctx, kind, from, to, geoms [, ownFrames, images, width, height]): from / to = F point sets each (3 or 4 points
 * x,y float32; the matrix of frame f maps from[f] -> to[f]: pass dst, src for the inverse warp, the solves run on the GPU like the
 * reference's :994 runs per warp) */
static napi_value fn_warp_inverse_geometric_batch(napi_env env, napi_callback_info info)
{
    napi_value a[9]; size_t argc = 9;
    if (napi_get_cb_info(env, info, &argc, a, NULL, NULL) != napi_ok || argc < 5) return throw_str(env, "hgwarp: wrong number of arguments");
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    size_t nf, nt, ng; batch_job job = { JOB_GEO_INV, 0, NULL, NULL, NULL, NULL, 0, 0, 0 };
    if (!get_i32(env, a[1], &job.kind)) return NULL;
    if (job.kind != HG_AFFINE && job.kind != HG_PROJECTIVE) return throw_str(env, "hgwarp: kind must be 0 (affine) or 1 (projective)");
    job.from = (const float *)get_typed(env, a[2], napi_float32_array, &nf, "fromPoints"); if (!job.from) return NULL;
    job.to = (const float *)get_typed(env, a[3], napi_float32_array, &nt, "toPoints"); if (!job.to) return NULL;
    int32_t *gv = (int32_t *)get_typed(env, a[4], napi_int32_array, &ng, "geoms"); if (!gv) return NULL;
    const int F = (int)(ng / 4);
    job.per = job.kind == HG_AFFINE ? 6 : 8;
    if (F <= 0) return throw_str(env, "hgwarp: geoms must hold 4 integers per frame");
    if (nf < (size_t)F * job.per || nt < (size_t)F * job.per) return throw_str(env, "hgwarp: point sets must hold frames x points x,y pairs");
    return run_batch(env, h, &job, gv, F, 0, a + 5, argc - 5, "warpInverseGeometricBatch");
}

/* warpForwardPiecewiseBatch(ctx, dstPoints, maxSrcX, maxSrcY, geoms [, ownFrames, images, width, height]): the frames warp() sends down the
 * FORWARD piecewise path (:421-422: output not larger than the input and at least input / 1.2) */
static napi_value fn_warp_forward_piecewise_batch(napi_env env, napi_callback_info info)
{
    napi_value a[9]; size_t argc = 9;
    if (napi_get_cb_info(env, info, &argc, a, NULL, NULL) != napi_ok || argc < 5) return throw_str(env, "hgwarp: wrong number of arguments");
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    size_t nd, ng; batch_job job = { JOB_PW_FWD, 0, NULL, NULL, NULL, NULL, 0, 0, 0 };
    job.dst = (const float *)get_typed(env, a[1], napi_float32_array, &nd, "dstPoints"); if (!job.dst) return NULL;
    if (!get_i32(env, a[2], &job.mx) || !get_i32(env, a[3], &job.my)) return NULL;
    int32_t *gv = (int32_t *)get_typed(env, a[4], napi_int32_array, &ng, "geoms"); if (!gv) return NULL;
    const int F = (int)(ng / 4);
    if (F <= 0) return throw_str(env, "hgwarp: geoms must hold 4 integers per frame");
    if (h->n_pts == 0 || nd < (size_t)F * 2 * h->n_pts) return throw_str(env, "hgwarp: dstPoints must hold frames x mesh points x,y pairs (piecewiseSetMesh first)");
    return run_batch(env, h, &job, gv, F, 1, a + 5, argc - 5, "warpForwardPiecewiseBatch");
}

/* warpForwardGeometricBatch(ctx, kind, mats Float64Array F x 8 (the FORWARD matrices, 6 used), geoms [, ownFrames, images, width, height]): the
 * affine frames warp() sends forward (:426-427: output of the source's size) */
static napi_value fn_warp_forward_geometric_batch(napi_env env, napi_callback_info info)
{
    napi_value a[8]; size_t argc = 8;
    if (napi_get_cb_info(env, info, &argc, a, NULL, NULL) != napi_ok || argc < 4) return throw_str(env, "hgwarp: wrong number of arguments");
    handle_t *h = get_handle(env, a[0]); if (!h) return NULL;
    size_t nm, ng; batch_job job = { JOB_GEO_FWD, 0, NULL, NULL, NULL, NULL, 0, 0, 0 };
    if (!get_i32(env, a[1], &job.kind)) return NULL;
    if (job.kind != HG_AFFINE && job.kind != HG_PROJECTIVE) return throw_str(env, "hgwarp: kind must be 0 (affine) or 1 (projective)");
    job.m = (const double *)get_typed(env, a[2], napi_float64_array, &nm, "matrices"); if (!job.m) return NULL;
    int32_t *gv = (int32_t *)get_typed(env, a[3], napi_int32_array, &ng, "geoms"); if (!gv) return NULL;
    const int F = (int)(ng / 4);
    if (F <= 0) return throw_str(env, "hgwarp: geoms must hold 4 integers per frame");
    if (nm < (size_t)F * 8) return throw_str(env, "hgwarp: matrices must hold 8 doubles per frame");
    return run_batch(env, h, &job, gv, F, 1, a + 4, argc - 4, "warpForwardGeometricBatch");
}

/* ---------------------------------------------------------------- several GPUs (hg_multi_*) */
typedef struct { hg_multi *m; size_t n_pts; } mhandle_t;

static void mhandle_finalize(napi_env env, void *data, void *hint)
{
    (void)env; (void)hint;
    mhandle_t *h = (mhandle_t *)data;
    if (h) { if (h->m) hg_multi_destroy(h->m); free(h); }
}

static mhandle_t *get_mhandle(napi_env env, napi_value v)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((mhandle_t *)p)->m) { throw_str(env, "hgwarp: invalid or destroyed multi-device handle"); return NULL; }
    return (mhandle_t *)p;
}

static napi_value throw_multi(napi_env env, hg_multi *m, const char *what, int code)
{
    char buf[512];
    snprintf(buf, sizeof buf, "hgwarp %s failed (%d): %s", what, code, hg_multi_last_error(m));
    return throw_str(env, buf);
}

static napi_value fn_multi_create(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    size_t n;
    int32_t *ids = (int32_t *)get_typed(env, a[0], napi_int32_array, &n, "devices"); if (!ids) return NULL;
    if (n == 0) return throw_str(env, "hgwarp: multiCreate needs at least one device id");
    mhandle_t *h = (mhandle_t *)calloc(1, sizeof *h);
    int rc = hg_multi_create((const int *)ids, (int)n, &h->m);
    if (rc != HG_OK) { free(h); return throw_multi(env, NULL, "hg_multi_create", rc); }
    napi_value ext;
    NAPI_OK(napi_create_external(env, h, mhandle_finalize, NULL, &ext));
    return ext;
}

static napi_value fn_multi_destroy(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    void *p = NULL;
    if (napi_get_value_external(env, a[0], &p) == napi_ok && p && ((mhandle_t *)p)->m) { hg_multi_destroy(((mhandle_t *)p)->m); ((mhandle_t *)p)->m = NULL; }
    return NULL;
}

static napi_value fn_multi_set_image(napi_env env, napi_callback_info info)
{
    napi_value a[4];
    if (!get_args(env, info, 4, a)) return NULL;
    mhandle_t *h = get_mhandle(env, a[0]); if (!h) return NULL;
    size_t n; int w, hh;
    uint8_t *px = (uint8_t *)get_typed(env, a[1], napi_uint8_clamped_array, &n, "image.data"); if (!px) return NULL;
    if (!get_i32(env, a[2], &w) || !get_i32(env, a[3], &hh)) return NULL;
    if (w <= 0 || hh <= 0 || n < (size_t)w * (size_t)hh * 4) return throw_str(env, "hgwarp: image.data is smaller than width*height*4");
    int rc = hg_multi_set_image(h->m, px, w, hh);
    if (rc != HG_OK) return throw_multi(env, h->m, "hg_multi_set_image", rc);
    return NULL;
}

static napi_value fn_multi_set_mesh(napi_env env, napi_callback_info info)
{
    napi_value a[5];
    if (!get_args(env, info, 5, a)) return NULL;
    mhandle_t *h = get_mhandle(env, a[0]); if (!h) return NULL;
    size_t np, nt; int msx, msy;
    float *src = (float *)get_typed(env, a[1], napi_float32_array, &np, "srcPoints"); if (!src) return NULL;
    uint32_t *tris = (uint32_t *)get_typed(env, a[2], napi_uint32_array, &nt, "triangles"); if (!tris) return NULL;
    if (!get_i32(env, a[3], &msx) || !get_i32(env, a[4], &msy)) return NULL;
    int rc = hg_multi_piecewise_set_mesh(h->m, src, (int)(np / 2), tris, (int)(nt / 3), msx, msy);
    if (rc != HG_OK) return throw_multi(env, h->m, "hg_multi_piecewise_set_mesh", rc);
    h->n_pts = np / 2;
    return NULL;
}

/* Optional per-frame sources of a multi-device batch: a JS array of n Uint8ClampedArray (width*height*4 each) -> F host pointers,
 * frame f reads images[f % n] (the rule of setImages).  Returns 1 with *out = NULL when `v` is undefined / null (shared source). */
static int get_frame_images(napi_env env, napi_value v, napi_value vw, napi_value vh, int F, const uint8_t ***out, int *w, int *hh)
{
    *out = NULL;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    if (t == napi_undefined || t == napi_null) return 1;
    bool is_arr = false; uint32_t n = 0;
    if (napi_is_array(env, v, &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, v, &n) != napi_ok || n == 0) {
        throw_str(env, "hgwarp: images must be a non-empty array of image data arrays"); return 0;
    }
    if (!get_i32(env, vw, w) || !get_i32(env, vh, hh)) return 0;
    if (*w <= 0 || *hh <= 0) { throw_str(env, "hgwarp: bad image size"); return 0; }
    const size_t bytes = (size_t)*w * (size_t)*hh * 4;
    const uint8_t **ptrs = (const uint8_t **)calloc((size_t)F, sizeof *ptrs);
    if (!ptrs) { throw_str(env, "hgwarp: out of memory"); return 0; }
    for (int f = 0; f < F; f++) {
        napi_value el; size_t len;
        if (napi_get_element(env, v, (uint32_t)f % n, &el) != napi_ok) { free(ptrs); throw_str(env, "hgwarp: N-API call failed"); return 0; }
        const uint8_t *px = (const uint8_t *)get_typed(env, el, napi_uint8_clamped_array, &len, "images[k]");
        if (!px) { free(ptrs); return 0; }
        if (len < bytes) { free(ptrs); throw_str(env, "hgwarp: an image is smaller than width*height*4"); return 0; }
        ptrs[f] = px;
    }
    *out = ptrs;
    return 1;
}

/* warpBatch over the device list: dst = F x 2N float32, geoms = Int32Array F x 4 [, images, width, height: one source per frame,
 * every device uploads its own block]; returns an Array of F Uint8ClampedArray
 * (pooled pinned memory where possible, so that the D2H copies of different devices run at the same time). */
static napi_value fn_multi_warp_batch(napi_env env, napi_callback_info info)
{
    napi_value a[6];
    size_t argc = 6;
    if (napi_get_cb_info(env, info, &argc, a, NULL, NULL) != napi_ok || argc < 3) return throw_str(env, "hgwarp: multiWarpBatch(multi, dstPoints, geoms[, images, width, height])");
    mhandle_t *h = get_mhandle(env, a[0]); if (!h) return NULL;
    size_t nd, ng;
    float *dst = (float *)get_typed(env, a[1], napi_float32_array, &nd, "dstPoints"); if (!dst) return NULL;
    int32_t *gv = (int32_t *)get_typed(env, a[2], napi_int32_array, &ng, "geoms"); if (!gv) return NULL;
    const int F = (int)(ng / 4);
    if (F <= 0) return throw_str(env, "hgwarp: geoms must hold 4 integers per frame");
    if (h->n_pts == 0 || nd < (size_t)F * 2 * h->n_pts) return throw_str(env, "hgwarp: dstPoints must hold frames x mesh points x,y pairs (multiSetMesh first)");
    napi_value arr;
    NAPI_OK(napi_create_array_with_length(env, F, &arr));
    uint8_t **outs = (uint8_t **)calloc((size_t)F, sizeof *outs);
    uint8_t dummy = 0;
    for (int f = 0; f < F; f++) {
        const hg_geom *g = (const hg_geom *)gv + f;
        const size_t px = (g->obj_w > 0 && g->obj_h > 0) ? (size_t)g->obj_w * g->obj_h : 0;
        void *out = NULL; napi_value ta = make_pixels(env, px * 4, NULL, 0, &out);
        if (!ta) { free(outs); return NULL; }
        outs[f] = px ? (uint8_t *)out : &dummy;
        napi_set_element(env, arr, f, ta);
    }
    const uint8_t **imgs = NULL; int iw = 0, ih = 0;
    if (argc >= 6 && !get_frame_images(env, a[3], a[4], a[5], F, &imgs, &iw, &ih)) { free(outs); return NULL; }
    int rc = imgs ? hg_multi_warp_piecewise_batch_images(h->m, dst, (const hg_geom *)gv, F, imgs, iw, ih, outs)
                  : hg_multi_warp_piecewise_batch(h->m, dst, (const hg_geom *)gv, F, outs);
    free(outs); free(imgs);
    if (rc != HG_OK) return throw_multi(env, h->m, imgs ? "hg_multi_warp_piecewise_batch_images" : "hg_multi_warp_piecewise_batch", rc);
    return arr;
}

static napi_value fn_multi_warp_geometric_batch(napi_env env, napi_callback_info info)
{
    napi_value a[8];
    size_t argc = 8;
    if (napi_get_cb_info(env, info, &argc, a, NULL, NULL) != napi_ok || argc < 5) return throw_str(env, "hgwarp: multiWarpGeometricBatch(multi, kind, from, to, geoms[, images, width, height])");
    mhandle_t *h = get_mhandle(env, a[0]); if (!h) return NULL;
    int kind; size_t nf, nt, ng;
    if (!get_i32(env, a[1], &kind)) return NULL;
    if (kind != HG_AFFINE && kind != HG_PROJECTIVE) return throw_str(env, "hgwarp: kind must be 0 (affine) or 1 (projective)");
    float *from = (float *)get_typed(env, a[2], napi_float32_array, &nf, "fromPoints"); if (!from) return NULL;
    float *to = (float *)get_typed(env, a[3], napi_float32_array, &nt, "toPoints"); if (!to) return NULL;
    int32_t *gv = (int32_t *)get_typed(env, a[4], napi_int32_array, &ng, "geoms"); if (!gv) return NULL;
    const int F = (int)(ng / 4);
    const size_t per = kind == HG_AFFINE ? 6 : 8;
    if (F <= 0) return throw_str(env, "hgwarp: geoms must hold 4 integers per frame");
    if (nf < (size_t)F * per || nt < (size_t)F * per) return throw_str(env, "hgwarp: point sets must hold frames x points x,y pairs");
    napi_value arr;
    NAPI_OK(napi_create_array_with_length(env, F, &arr));
    uint8_t **outs = (uint8_t **)calloc((size_t)F, sizeof *outs);
    uint8_t dummy = 0;
    for (int f = 0; f < F; f++) {
        const hg_geom *g = (const hg_geom *)gv + f;
        const size_t px = (g->obj_w > 0 && g->obj_h > 0) ? (size_t)g->obj_w * g->obj_h : 0;
        void *out = NULL; napi_value ta = make_pixels(env, px * 4, NULL, 0, &out);
        if (!ta) { free(outs); return NULL; }
        outs[f] = px ? (uint8_t *)out : &dummy;
        napi_set_element(env, arr, f, ta);
    }
    const uint8_t **imgs = NULL; int iw = 0, ih = 0;
    if (argc >= 8 && !get_frame_images(env, a[5], a[6], a[7], F, &imgs, &iw, &ih)) { free(outs); return NULL; }
    int rc = imgs ? hg_multi_warp_geometric_batch_images(h->m, kind, from, to, (const hg_geom *)gv, F, imgs, iw, ih, outs)
                  : hg_multi_warp_geometric_batch(h->m, kind, from, to, (const hg_geom *)gv, F, outs);
    free(outs); free(imgs);
    if (rc != HG_OK) return throw_multi(env, h->m, imgs ? "hg_multi_warp_geometric_batch_images" : "hg_multi_warp_geometric_batch", rc);
    return arr;
}

/* ---------------------------------------------------------------- module */
static void pool_free(napi_env env, void *data, void *hint)
{
    (void)env; (void)hint;
    pool_t *P = (pool_t *)data;
    if (!P) return;
    /* the env is gone and every ArrayBuffer with it: the buffers go back to the driver */
    for (int i = 0; i < P->npins; i++) if (P->pins[i].ptr) { if (P->pool_malloc) free(P->pins[i].ptr); else hg_host_free(P->pins[i].ptr); }
    free(P);
}

static napi_value init(napi_env env, napi_value exports)
{
    pool_t *P = (pool_t *)calloc(1, sizeof *P);
    if (!P) return NULL;
    P->pin_limit = (size_t)2 << 30;
    const napi_node_version *nv = NULL;
    P->early_reuse = (napi_get_node_version(env, &nv) == napi_ok && nv && nv->major <= 12) ? 1 : 0;
    if (getenv("HGWARP_POOL_NO_EARLY_REUSE")) P->early_reuse = 0;           /* (tests: the V8 >= 8 life cycle under Node 12) */
    if (napi_set_instance_data(env, P, pool_free, NULL) != napi_ok) { free(P); return NULL; }
    static const struct { const char *name; napi_callback fn; } fns[] = {
        { "create", fn_create }, { "destroy", fn_destroy }, { "deviceCount", fn_device_count },
        { "solveAffine", fn_solve_affine }, { "invertAffine", fn_invert_affine }, { "solveProjective", fn_solve_projective },
        { "transformLimits", fn_transform_limits }, { "minmaxXY", fn_minmax_xy }, { "triangulate", fn_triangulate },
        { "setImage", fn_set_image }, { "setImages", fn_set_images }, { "warpInverseGeometric", fn_warp_inverse_geometric },
        { "piecewiseSetMesh", fn_piecewise_set_mesh }, { "piecewisePrepare", fn_piecewise_prepare },
        { "warpInversePiecewise", fn_warp_inverse_piecewise }, { "getTriMap", fn_get_tri_map }, { "getMatrices", fn_get_matrices },
        { "warpInversePiecewiseBatch", fn_warp_inverse_piecewise_batch }, { "warpInverseGeometricBatch", fn_warp_inverse_geometric_batch },
        { "warpForwardGeometric", fn_warp_forward_geometric }, { "warpForwardPiecewise", fn_warp_forward_piecewise },
        { "warpForwardPiecewiseBatch", fn_warp_forward_piecewise_batch }, { "warpForwardGeometricBatch", fn_warp_forward_geometric_batch },
        { "releaseBatch", fn_release_batch }, { "pinnedBuffer", fn_pinned_buffer },
        { "solveAffineTriangles", fn_solve_affine_triangles }, { "warpInversePiecewiseState", fn_warp_inverse_piecewise_state },
        { "warpForwardPiecewiseState", fn_warp_forward_piecewise_state },
        { "release", fn_release }, { "setPinnedLimit", fn_set_pinned_limit }, { "poolStats", fn_pool_stats }, { "redoneFrames", fn_redone_frames }, { "_poolTestFrames", fn_pool_test_frames }, { "poolPressure", fn_pool_pressure }, { "poolCollected", fn_pool_collected },
        { "multiCreate", fn_multi_create }, { "multiDestroy", fn_multi_destroy }, { "multiSetImage", fn_multi_set_image },
        { "multiSetMesh", fn_multi_set_mesh }, { "multiWarpBatch", fn_multi_warp_batch }, { "multiWarpGeometricBatch", fn_multi_warp_geometric_batch },
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
