// hg_api.hip -- the C ABI of include/hgwarp.h, part 1: library / context / buffer management, options, timing, host-side
// solves, the source image.  (hg_api_geometric.hip, hg_api_piecewise.hip, hg_api_forward.hip hold the warp entry points.)
// No torch, no C++ types in the exported signatures.  There is deliberately no CPU warp path in these files.
#include "hg_ctx.h"

thread_local std::string g_err;

// ------------------------------------------------------------------------------------------------ library / context
extern "C" int hg_version(void) { return HG_VERSION; }

extern "C" int hg_device_count(int *count)
{
    if (!count) return fail(nullptr, HG_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(nullptr, HG_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    return HG_OK;
}

static int create_common(int device_id, void *stream, bool own, hg_ctx **out)
{
    if (!out) return fail(nullptr, HG_ERR_INVALID, "ctx out pointer is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, HG_ERR_NO_DEVICE, std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "0 devices") +
                    "); libhgwarp has no CPU fallback");
    if (device_id < 0 || device_id >= n) return fail(nullptr, HG_ERR_NO_DEVICE, "device id out of range");
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, HG_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + "; libhgwarp is built for gfx950 (MI355X) only");
    hg_ctx *c = new hg_ctx();
    c->device = device_id;
    {   // XCC count of this device / partition (8 on an unpartitioned MI355X): the warp kernels map block ids to per-XCD row bands
        // with it (speed only -- any value gives the same pixels).  Not a power of two or unknown: no banding.
        int nx = 0;
        if (hipDeviceGetAttribute(&nx, hipDeviceAttributeNumberOfXccs, device_id) != hipSuccess) nx = 1;
        c->xcc_log2 = 0;
        if (nx > 0 && (nx & (nx - 1)) == 0) while ((1 << c->xcc_log2) < nx && c->xcc_log2 < 6) c->xcc_log2++;
    }
    if (hipSetDevice(device_id) != hipSuccess) { delete c; return fail(nullptr, HG_ERR_HIP, "hipSetDevice failed"); }
    if (own) {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(nullptr, HG_ERR_HIP, "hipStreamCreate failed"); }
        c->own_stream = true;
    } else {
        c->stream = static_cast<hipStream_t>(stream);
    }
    *out = c;
    return HG_OK;
}

extern "C" int hg_create(int device_id, hg_ctx **ctx) { return create_common(device_id, nullptr, true, ctx); }
extern "C" int hg_create_on_stream(int device_id, void *hip_stream, hg_ctx **ctx) { return create_common(device_id, hip_stream, false, ctx); }

extern "C" void hg_destroy(hg_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream || !c->own_stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_img && !c->img_aliased) (void)hipFree(c->d_img);
    void *ptrs[] = { c->d_src, c->d_tris, c->d_set, c->d_trir, c->d_trix, c->d_segs, c->d_fwd, c->d_inv, c->d_status, c->d_two_round, c->d_rowcnt, c->d_rowent, c->d_bands,
                     c->d_geo_frames, c->d_mats, c->d_geo_pts, c->d_geo_plain, c->d_map32, c->d_fmap, c->d_win32, c->d_fwd_par, c->d_fbbox, c->d_frowoff, c->d_frowext, c->d_ftile_cnt, c->d_fwd_status, c->d_ftile_ent, c->d_map16, c->d_out_tmp };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    for (hg_ctx::Stage &st : c->stage) { if (st.h) (void)hipHostFree(st.h); if (st.done) (void)hipEventDestroy(st.done); }
    for (hg_ctx::GeoStage &gs : c->geo_stage) { if (gs.h) (void)hipHostFree(gs.h); if (gs.done) (void)hipEventDestroy(gs.done); }
    { void *rp[] = { c->d_redo_frame, c->d_redo_dst, c->d_redo_trir, c->d_redo_trix, c->d_redo_segs, c->d_redo_fwd, c->d_redo_inv, c->d_redo_status, c->d_st_pts, c->d_st_tris, c->d_st_mats };
      for (void *q : rp) if (q) (void)hipFree(q); }
    for (int i = 0; i < hg_ctx::kEvRing; i++) { if (c->ev0[i]) (void)hipEventDestroy(c->ev0[i]); if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]); }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->copy_event) (void)hipEventDestroy(c->copy_event);
    if (c->down_stream) { (void)hipStreamSynchronize(c->down_stream); (void)hipStreamDestroy(c->down_stream); }
    if (c->down_event) (void)hipEventDestroy(c->down_event);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char *hg_last_error(const hg_ctx *c) { return c ? c->err.c_str() : g_err.c_str(); }

extern "C" int hg_device_alloc(hg_ctx *c, size_t bytes, void **dptr)
{
    HG_TRY(bind(c));
    if (!dptr) return fail(c, HG_ERR_INVALID, "dptr is NULL");
    *dptr = nullptr;
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(c, HG_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return HG_OK;
}

extern "C" int hg_device_free(hg_ctx *c, void *dptr)
{
    HG_TRY(bind(c));
    // Queued runs are settled first: a frame one of them flagged is redone INTO its output buffer by hg_sync, which must not
    // happen after that buffer has been freed.
    HG_TRY(hg_sync(c));
    if (dptr) HIP_TRY(c, hipFree(dptr));
    return HG_OK;
}

extern "C" int hg_copy_to_host(hg_ctx *c, void *dst, const void *src, size_t bytes)
{
    HG_TRY(bind(c));
    if (!dst || !src) return fail(c, HG_ERR_INVALID, "NULL pointer");
    HG_TRY(hg_sync(c));
    HIP_TRY(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return HG_OK;
}

// D2H on the ctx stream, after everything queued so far; the caller keeps dst alive until hg_sync().  Settles queued
// piecewise runs first (frames the fused path flagged are redone before they are copied).  With pinned destination memory
// (hg_host_alloc) the copy is a true asynchronous DMA, so several devices / frames overlap.
extern "C" int hg_copy_to_host_async(hg_ctx *c, void *dst, const void *src, size_t bytes)
{
    HG_TRY(bind(c));
    if (!dst || !src) return fail(c, HG_ERR_INVALID, "NULL pointer");
    if (!c->pw_pending_out.empty() || !c->fwd_pending.empty()) HG_TRY(hg_sync(c));
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return HG_OK;
}

// The same copy WITHOUT settling queued runs first: purely stream-ordered.  For callers that queue the copies of several devices
// before waiting for any of them (hg_multi_*): frames a fused run only flagged are rewritten by the later hg_sync, after which
// the caller copies them again (hg_redone_frames tells).
extern "C" int hg_enqueue_copy_to_host(hg_ctx *c, void *dst, const void *src, size_t bytes)
{
    HG_TRY(bind(c));
    if (!dst || !src) return fail(c, HG_ERR_INVALID, "NULL pointer");
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return HG_OK;
}

// Host -> device on the context's COPY stream (a second stream, created on first use): not ordered with the warp stream until
// hg_fence_copies().  Lets a binding upload the source of frame f + 1 while frame f's pixels travel the other way on the warp stream
// (PCIe is full duplex, the two directions have their own DMA engines).  Pageable `src` is staged by the runtime: the call then
// returns once the caller's memory has been read (which is exactly what keeps the host busy while the opposite copy runs).
extern "C" int hg_upload_on_copy_stream(hg_ctx *c, void *dst_device, const void *src_host, size_t bytes)
{
    HG_TRY(bind(c));
    if (!dst_device || !src_host) return fail(c, HG_ERR_INVALID, "NULL pointer");
    if (!c->copy_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, c->copy_stream));
    return HG_OK;
}

// Everything queued on the warp stream after this call waits for the uploads queued on the copy stream so far.
extern "C" int hg_fence_copies(hg_ctx *c)
{
    HG_TRY(bind(c));
    if (!c->copy_stream) return HG_OK;
    if (!c->copy_event) HIP_TRY(c, hipEventCreateWithFlags(&c->copy_event, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->copy_event, c->copy_stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->copy_event, 0));
    return HG_OK;
}

// D2H on the context's DOWNLOAD stream, ordered behind everything issued on the warp stream so far -- and nothing issued later: frame f
// comes down while frame f + 1 is warped (and while image f + 2 goes up on the copy stream).  Purely stream-ordered like
// hg_enqueue_copy_to_host: a frame the fused run only flagged is rewritten by a later hg_sync (hg_redone_frames tells) and must be
// downloaded again.  hg_fence_downloads: the host waits for the download stream (hg_sync does not).
extern "C" int hg_download_behind_warps(hg_ctx *c, void *dst, const void *src, size_t bytes)
{
    HG_TRY(bind(c));
    if (!dst || !src) return fail(c, HG_ERR_INVALID, "NULL pointer");
    if (!c->down_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
    if (!c->down_event) HIP_TRY(c, hipEventCreateWithFlags(&c->down_event, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->down_event, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->down_stream, c->down_event, 0));
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->down_stream));
    return HG_OK;
}

extern "C" int hg_fence_downloads(hg_ctx *c)
{
    HG_TRY(bind(c));
    if (c->down_stream) HIP_TRY(c, hipStreamSynchronize(c->down_stream));
    return HG_OK;
}

// Everything queued on the ctx stream after this call waits for `hip_event` (a hipEvent_t recorded on any stream of any
// device): lets a caller order warps behind its own uploads / peer copies without blocking the host.
extern "C" int hg_stream_wait_event(hg_ctx *c, void *hip_event)
{
    HG_TRY(bind(c));
    if (!hip_event) return fail(c, HG_ERR_INVALID, "event is NULL");
    HIP_TRY(c, hipStreamWaitEvent(c->stream, static_cast<hipEvent_t>(hip_event), 0));
    return HG_OK;
}

// Pinned (page-locked, portable across devices) host memory for frames that leave the GPU: DMA at full PCIe rate without
// the runtime's staging copy, and no first-touch page faults once the buffer is being reused.
extern "C" int hg_host_alloc(size_t bytes, void **p)
{
    if (!p) return fail(nullptr, HG_ERR_INVALID, "p is NULL");
    *p = nullptr;
    hipError_t e = hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) return fail(nullptr, HG_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    return HG_OK;
}

extern "C" int hg_host_free(void *p)
{
    if (p && hipHostFree(p) != hipSuccess) return fail(nullptr, HG_ERR_HIP, "hipHostFree failed");
    return HG_OK;
}

extern "C" int hg_ctx_device(const hg_ctx *c) { return c ? c->device : -1; }

extern "C" int hg_copy_to_device(hg_ctx *c, void *dst, const void *src, size_t bytes)
{
    HG_TRY(bind(c));
    if (!dst || !src) return fail(c, HG_ERR_INVALID, "NULL pointer");
    HG_TRY(hg_sync(c));                                  // queued warps may still read the destination
    HIP_TRY(c, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return HG_OK;
}

extern "C" int hg_last_piecewise_kernel(hg_ctx *c) { return c ? c->pw_last_kernel : 0; }
extern "C" int hg_last_piecewise_variant(hg_ctx *c) { return c ? c->pw_last_variant : 0; }
extern "C" int hg_last_forward_kernel(hg_ctx *c) { return c ? c->fwd_last_kernel : 0; }

extern "C" int hg_last_piecewise_self(hg_ctx *c) { return c && c->pw_self ? 1 : 0; }
extern "C" int hg_last_piecewise_flag(hg_ctx *c) { return c ? c->pw_last_flag : 0; }
extern "C" long hg_redone_frames(hg_ctx *c) { return c ? c->pw_redone : 0; }
extern "C" long hg_layout_walks(hg_ctx *c) { return c ? c->pw_layout_walks : 0; }

extern "C" int hg_set_option(hg_ctx *c, const char *key, int value)
{
    HG_TRY(bind(c));
    if (!key) return fail(c, HG_ERR_INVALID, "hg_set_option: key is NULL");
    if (!std::strcmp(key, "min_row_groups")) c->opt_min_row_groups = value;
    else if (!std::strcmp(key, "patch")) c->opt_patch = value;
    else if (!std::strcmp(key, "phase")) c->opt_phase = value;
    else if (!std::strcmp(key, "geo_windows")) c->opt_geo_nw = value;
    else if (!std::strcmp(key, "hi_bounds")) c->opt_hi_bounds = value != 0;
    else if (!std::strcmp(key, "xcc_rotate")) c->opt_xcc_rotate = value;
    else if (!std::strcmp(key, "sub_bands")) c->opt_sub_bands = std::min(value, 64);
    else if (!std::strcmp(key, "compact")) c->opt_compact = value < 0 ? -1 : (value ? 1 : 0);
    else if (!std::strcmp(key, "tile")) { c->opt_tile = value < 0 ? -1 : (value ? 1 : 0); c->pw_tile_disabled = false; }
    else if (!std::strcmp(key, "self_spans")) { c->opt_self = value < 0 ? -1 : (value ? 1 : 0); c->pw_self_disabled = false; }
    else if (!std::strcmp(key, "tri_group")) c->opt_tri_group = value < 0 ? -1 : (value >= 64 ? 64 : (value ? 16 : 0));
    else if (!std::strcmp(key, "safe_spans")) c->opt_safe_spans = value < 0 ? -1 : (value ? 1 : 0);
    else if (!std::strcmp(key, "upload_kernel")) c->opt_upload_kernel = value < 0 ? -1 : (value ? 1 : 0);
    else if (!std::strcmp(key, "xcc")) {                      // block id -> XCD band mapping for `value` XCCs (a power of two <= 64); speed only
        if (value < 1 || value > 64 || (value & (value - 1))) return fail(c, HG_ERR_INVALID, "hg_set_option: xcc must be a power of two in 1..64");
        c->xcc_log2 = 0;
        while ((1 << c->xcc_log2) < value) c->xcc_log2++;
    }
    else if (!std::strcmp(key, "fwd_tiles")) { c->opt_fwd_tiles = value; c->fwd_pw_tiles_disabled = false; }
    else return fail(c, HG_ERR_INVALID, std::string("hg_set_option: unknown key ") + key);
    return HG_OK;
}

extern "C" int hg_xcc_count(const hg_ctx *c) { return c ? 1 << c->xcc_log2 : 0; }

extern "C" int hg_set_timing(hg_ctx *c, int enabled)
{
    HG_TRY(bind(c));
    c->timing = enabled != 0;
    c->ev_count = 0;
    if (c->timing && !c->ev0[0]) {
        for (int i = 0; i < hg_ctx::kEvRing; i++) { HIP_TRY(c, hipEventCreate(&c->ev0[i])); HIP_TRY(c, hipEventCreate(&c->ev1[i])); }
    }
    return HG_OK;
}

int time_begin(hg_ctx *c)
{
    if (c->timing) HIP_TRY(c, hipEventRecord(c->ev0[c->ev_count % hg_ctx::kEvRing], c->stream));
    return HG_OK;
}

int time_end(hg_ctx *c)
{
    if (c->timing) { HIP_TRY(c, hipEventRecord(c->ev1[c->ev_count % hg_ctx::kEvRing], c->stream)); c->ev_count++; }
    return HG_OK;
}

extern "C" int hg_kernel_ms_stats(hg_ctx *c, double *total_ms, int *launches)
{
    HG_TRY(bind(c));
    if (!total_ms || !launches) return fail(c, HG_ERR_INVALID, "NULL pointer");
    *total_ms = 0.0; *launches = 0;
    if (!c->timing || c->ev_count == 0) return HG_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const long n = std::min<long>(c->ev_count, hg_ctx::kEvRing);
    for (long k = 0; k < n; k++) {
        const int i = (int)((c->ev_count - 1 - k) % hg_ctx::kEvRing);
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]));
        *total_ms += ms;
    }
    *launches = (int)n;
    return HG_OK;
}

extern "C" int hg_last_kernel_ms(hg_ctx *c, float *ms)
{
    HG_TRY(bind(c));
    if (!ms) return fail(c, HG_ERR_INVALID, "ms is NULL");
    *ms = 0.f;
    if (!c->timing || c->ev_count == 0) return HG_OK;
    const int i = (int)((c->ev_count - 1) % hg_ctx::kEvRing);
    HIP_TRY(c, hipEventSynchronize(c->ev1[i]));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev0[i], c->ev1[i]));
    return HG_OK;
}

// ------------------------------------------------------------------------------------------------ host-side solves
extern "C" double hg_js_round(double x) { return js_round(x); }

extern "C" int hg_solve_affine(const float src[6], const float dst[6], float out[6])
{
    if (!src || !dst || !out) return fail(nullptr, HG_ERR_INVALID, "NULL pointer");
    solve_affine(src, dst, out);
    return HG_OK;
}

extern "C" int hg_invert_affine(const float m[6], float out[6])
{
    if (!m || !out) return fail(nullptr, HG_ERR_INVALID, "NULL pointer");
    invert_affine(m, out);
    return HG_OK;
}

// projectiveMatrixFromSquares :1320-1333 + numeric.js solve :1650-1751.  One restatement for host and device
// (solve_projective_regs, hg_math.h): the same function k_solve_frames runs per frame on the GPU.
extern "C" int hg_solve_projective(const float s[8], const float d[8], double out[8])
{
    if (!s || !d || !out) return fail(nullptr, HG_ERR_INVALID, "NULL pointer");
    solve_projective_regs(s, d, out);
    return HG_OK;
}

extern "C" int hg_transform_limits(int kind, const double *m, double w, double h, double out[4])
{
    if (!m || !out || (kind != HG_AFFINE && kind != HG_PROJECTIVE)) return fail(nullptr, HG_ERR_INVALID, "bad arguments");
    double px[4], py[4];
    const double cx[4] = { 0, 0, w, w }, cy[4] = { 0, h, 0, h };        // p0_0, p1_0, p0_1, p1_1 (:1506-1517)
    for (int i = 0; i < 4; i++) {
        if (kind == HG_AFFINE) apply_affine(m, cx[i], cy[i], px[i], py[i]); else apply_projective(m, cx[i], cy[i], px[i], py[i]);
    }
    auto mn4 = [](const double *v) { return js_min2(js_min2(v[0], v[1]), js_min2(v[2], v[3])); };
    auto mx4 = [](const double *v) { return js_max2(js_max2(v[0], v[1]), js_max2(v[2], v[3])); };
    const double xo = mn4(px), yo = mn4(py);
    out[0] = js_round(xo); out[1] = js_round(yo);                        // :1525
    out[2] = js_round(mx4(px) - xo); out[3] = js_round(mx4(py) - yo);
    return HG_OK;
}

extern "C" int hg_minmax_xy(const float *p, int n, double out[4])
{
    if ((!p && n > 0) || !out || n < 0) return fail(nullptr, HG_ERR_INVALID, "bad arguments");
    double maxX = -INFINITY, maxY = -INFINITY, minX = INFINITY, minY = INFINITY;
    for (int i = 0; i < n; i++) {
        const double e = p[i];
        if ((i & 1) == 0) { if (e > maxX) maxX = e; if (e < minX) minX = e; }
        else              { if (e > maxY) maxY = e; if (e < minY) minY = e; }
    }
    out[0] = js_round(minX); out[1] = js_round(minY); out[2] = js_round(maxX); out[3] = js_round(maxY);
    return HG_OK;
}

extern "C" int hg_pack_offsets(const hg_geom *g, int n, size_t *offsets, size_t *total)
{
    if ((!g && n > 0) || n < 0) return fail(nullptr, HG_ERR_INVALID, "bad arguments");
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        if (offsets) offsets[i] = off;
        const size_t bytes = (g[i].obj_w > 0 && g[i].obj_h > 0) ? (size_t)g[i].obj_w * (size_t)g[i].obj_h * 4 : 0;
        off += (bytes + 255) & ~(size_t)255;
    }
    if (total) *total = off;
    return HG_OK;
}

// ------------------------------------------------------------------------------------------------ source image
extern "C" int hg_set_image(hg_ctx *c, const uint8_t *rgba, int w, int h)
{
    HG_TRY(bind(c));
    if (!rgba || w <= 0 || h <= 0) return fail(c, HG_ERR_INVALID, "hg_set_image: bad image");
    HG_TRY(hg_sync(c));                                 // settle queued runs before their source is replaced
    const size_t bytes = (size_t)w * h * 4;
    if (c->img_aliased) { c->d_img = nullptr; c->img_cap = 0; c->img_aliased = false; }
    HG_TRY(ensure(c, c->d_img, c->img_cap, bytes));
    HIP_TRY(c, hipMemcpyAsync(c->d_img, rgba, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));       // caller memory is not retained after return
    c->W = w; c->H = h; c->n_imgs = 1; c->img_stride = 0;
    return HG_OK;
}

extern "C" int hg_set_image_device(hg_ctx *c, const void *d_rgba, int w, int h)
{
    HG_TRY(bind(c));
    if (!d_rgba || w <= 0 || h <= 0) return fail(c, HG_ERR_INVALID, "hg_set_image_device: bad image");
    HG_TRY(hg_sync(c));
    if (c->d_img && !c->img_aliased) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(c->d_img)); }
    c->d_img = const_cast<uint8_t *>(static_cast<const uint8_t *>(d_rgba));
    c->img_cap = 0; c->img_aliased = true;
    c->W = w; c->H = h; c->n_imgs = 1; c->img_stride = 0;
    return HG_OK;
}

extern "C" int hg_set_images_device(hg_ctx *c, const void *d_rgba, int w, int h, int n_images, size_t stride_bytes)
{
    if (n_images <= 0 || (w > 0 && h > 0 && n_images > 1 && (stride_bytes < (size_t)w * h * 4 || (stride_bytes & 3))))
        return fail(c, HG_ERR_INVALID, "hg_set_images_device: n_images must be >= 1 and the stride a multiple of 4 bytes >= width*height*4");
    HG_TRY(hg_set_image_device(c, d_rgba, w, h));
    c->n_imgs = n_images; c->img_stride = n_images > 1 ? stride_bytes : 0;
    return HG_OK;
}

// ------------------------------------------------------------------------------------------------ frames helpers
void output_layout(const std::vector<FrameDesc> &frames, size_t *extent, uint64_t *layout)
{
    size_t ext = 0;
    uint64_t h = 1469598103934665603ull;                     // FNV-1a over (offset, bytes) of every frame with pixels
    for (const FrameDesc &d : frames) {
        if (d.obj_w <= 0 || d.obj_h <= 0) continue;
        const uint64_t bytes = (uint64_t)d.obj_w * d.obj_h * 4, v[2] = { d.out_off, bytes };
        ext = std::max(ext, (size_t)(d.out_off + bytes));
        for (uint64_t x : v) for (int k = 0; k < 8; k++) { h ^= (x >> (8 * k)) & 0xff; h *= 1099511628211ull; }
    }
    *extent = ext; *layout = h ? h : 1;
}

int settle_output_conflicts(hg_ctx *c, const void *out, size_t extent, uint64_t layout)
{
    const uint8_t *lo = static_cast<const uint8_t *>(out), *hi = lo + extent;
    bool conflict = false;
    for (const hg_ctx::Pending &p : c->pw_pending_out)
        if (p.out < hi && lo < p.out + p.extent && !(layout != 0 && p.out == lo && p.extent == extent && p.layout == layout)) conflict = true;
    for (const hg_ctx::FwdPending &p : c->fwd_pending)
        if (p.out < hi && lo < p.out + p.extent && !(layout != 0 && p.out == lo && p.extent == extent && p.layout == layout)) conflict = true;
    return conflict ? hg_sync(c) : HG_OK;
}

int fill_frames(hg_ctx *c, std::vector<FrameDesc> &v, const hg_geom *geoms, const size_t *offs, int n)
{
    if (n > 65535) return fail(c, HG_ERR_INVALID, "more than 65535 frames in one set (the frame index is a grid dimension)");
    v.resize(n);
    size_t off = 0, moff = 0;
    for (int i = 0; i < n; i++) {
        FrameDesc &d = v[i];
        d.x_off = geoms[i].x_off; d.y_off = geoms[i].y_off; d.obj_w = geoms[i].obj_w; d.obj_h = geoms[i].obj_h;
        const size_t px = (d.obj_w > 0 && d.obj_h > 0) ? (size_t)d.obj_w * (size_t)d.obj_h : 0;
        if (px > ((size_t)1 << 31)) return fail(c, HG_ERR_INVALID, "frame larger than 2^31 pixels");
        // pixel coordinates x = xOff + column stay exact integers in every kernel (int32 sums, f32-matrix * x products in fp64)
        if (std::abs((int64_t)d.x_off) > (1 << 26) || std::abs((int64_t)d.y_off) > (1 << 26))
            return fail(c, HG_ERR_INVALID, "output window offset beyond 2^26 pixels");
        d.out_off = offs ? offs[i] : off;
        if (d.out_off & 3) return fail(c, HG_ERR_INVALID, "output offsets must be multiples of 4 bytes");
        d.map_off = moff;
        off += (px * 4 + 255) & ~(size_t)255;
        moff += px;
    }
    return HG_OK;
}

