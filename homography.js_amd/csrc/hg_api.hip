// hg_api.hip -- the C ABI of include/hgwarp.h: context / buffer management, host-side solves, kernel orchestration.
// No torch, no C++ types in the exported signatures.  There is deliberately no CPU warp path in this file.
#include "../../include/hgwarp.h"
#include "hg_kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace hg;

constexpr size_t kFwdStatusRing = 16;   // tile-binned forward piecewise batches that may be queued before their status words are checked
constexpr size_t kStatusRing = 64;      // fused piecewise runs that may be queued before their status words are checked

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;

struct hg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    int deferred = HG_OK;

    // source image
    uint8_t *d_img = nullptr; size_t img_cap = 0; bool img_aliased = false;
    int W = 0, H = 0;
    int n_imgs = 1; size_t img_stride = 0;                     // hg_set_images_device: frame f reads image f % n_imgs

    // mesh (source side)
    float *d_src = nullptr; size_t src_cap = 0;
    uint32_t *d_tris = nullptr; size_t tris_cap = 0;
    std::vector<uint32_t> h_tris;                              // host copies (row-density / shear estimates in hg_piecewise_set_frames)
    std::vector<float> h_src;
    int n_pts = 0, n_tris = 0, min_src_x = 0, min_src_y = 0;
    bool have_mesh = false;

    // piecewise frames
    std::vector<FrameDesc> pw_frames;          // host copy
    uint8_t *d_set = nullptr; size_t set_cap = 0;             // the frame set in ONE block: F frame records, then F x n_pts x 2 destiny floats (one upload)
    FrameDesc *d_pw_frames = nullptr;                         // = d_set
    float *d_dst = nullptr;                                   // = d_set + F * sizeof(FrameDesc)
    TriRange *d_trir = nullptr; size_t trir_cap = 0;
    Seg *d_segs = nullptr; size_t segs_cap = 0;
    float *d_fwd = nullptr; size_t fwd_cap = 0;
    float *d_inv = nullptr; size_t inv_cap = 0;
    int32_t *d_status = nullptr; size_t status_cap = 0;
    int32_t *status_ptr = nullptr;                             // where this frame set's status words live (d_status, or the tail of d_rowcnt)
    int32_t *h_status = nullptr; size_t h_status_cap = 0;      // pinned
    bool pw_setup_done = false;                                // the per-triangle solves ran for the uploaded frames
    // fast path: per-output-row span lists
    int32_t *d_rowcnt = nullptr; size_t rowcnt_cap = 0;
    uint8_t *d_rowent = nullptr; size_t rowent_cap = 0;        // bytes
    int pw_cover = 0;                                          // estimated longest per-row span list of the uploaded frames
    bool pw_compact = false;                                   // span lists use 8-byte entries (dense rows / k_pw_patch), else 32-byte
    int row_cap = 64;                                          // entries per row; grows (sticky) after an overflow
    bool pw_fast = false;                                      // uploaded frames are eligible for k_tri_spans/k_pw_rows
    bool rows_clean = false;                                   // span counters + the next status set were zeroed by the last k_pw_rows
    int status_slot = 0;                                       // which of the kStatusRing status-word sets the current step uses
    int32_t *status_base = nullptr, *status_next = nullptr;
    int pw_row_group = kRowGroup;                              // output rows per k_pw_rows workgroup (4, or 1 for dense meshes)
    int pw_tri_threads = 128;                                  // k_tri_spans workgroup size
    bool pw_patch = false;                                     // dense mesh that fits k_pw_patch (4-row groups, 2-D gather patches)
    bool pw_patch_fits = false;                                // ... the frame set is within k_pw_patch's limits (it may be preferred later: one source per frame)
    double pw_fill = 1.0;                                      // heaviest XCD row band / mean band (span counts per row), 1 = even rows
    int opt_xcc_rotate = -1;                                   // -1 by estimate, 0 / 1
    double pw_shear = 0.0;                                     // mean |d(source row) / d(output x)| of the uploaded frames (layout heuristic)
    bool pw_patch_dense = false;                               // ... only in its global-record variant (up to 511 spans per row)
    bool pw_patch_disabled = false;                            // a group exceeded k_pw_patch's limits once: stay with k_pw_rows
    bool pw_used_patch = false;                                // the last fused run went through k_pw_patch
    int pw_last_kernel = 0;                                    // hg_last_piecewise_kernel()
    long pw_redone = 0;                                        // frames redone through the materialised map (hg_redone_frames())
    int opt_min_row_groups = 1536, opt_patch = -1, opt_phase = -1, opt_geo_nw = 8;   // hg_set_option()
    int xcc_log2 = 3;                                          // log2(XCCs of the device): hipDeviceAttributeNumberOfXccs at hg_create, option "xcc"
    int opt_lds_pad = -1;                                       // KB of dynamic LDS padding per k_pw_rows workgroup (occupancy experiments)
    int opt_sgpr_cap = -1;                                     // -1 auto (shared source), 0 never, 1 always: k_pw_rows_s80
    int opt_hi_bounds = 1;                                     // 0: fp64 bounds compares instead of the high-dword form (hg_dev.h)
    // fused runs whose per-frame status words have not been checked yet: up to kStatusRing - 1 calls are queued back to back
    // with nothing but their two kernels in the stream; each flags into its own set of status words, read back by hg_sync
    struct Pending { uint8_t *out; int slot; int stage; };     // stage: which staged frame set (points + windows) the run warped
    std::vector<Pending> pw_pending_out;
    // Frame sets arrive through a ring of page-locked staging buffers (FrameDesc[F], then the F x n_pts x 2 destination
    // points): hg_piecewise_set_frames copies the caller's arrays there and queues stream-ordered uploads -- it neither waits
    // for the GPU nor keeps caller memory.  A staged set stays intact until every run that used it has been settled, so frames a
    // fused run flagged can still be redone (through the materialised map) after newer sets were uploaded.
    struct Stage { uint8_t *h = nullptr; size_t cap = 0; int n = 0, n_pts = 0; };
    Stage stage[kStatusRing];
    int stage_cur = -1;
    // scratch of the deferred redo (one frame): its FrameDesc, points, solves
    FrameDesc *d_redo_frame = nullptr; size_t redo_frame_cap = 0;
    float *d_redo_dst = nullptr; size_t redo_dst_cap = 0;
    TriRange *d_redo_trir = nullptr; size_t redo_trir_cap = 0;
    Seg *d_redo_segs = nullptr; size_t redo_segs_cap = 0;
    float *d_redo_fwd = nullptr; size_t redo_fwd_cap = 0;
    float *d_redo_inv = nullptr; size_t redo_inv_cap = 0;
    int32_t *d_redo_status = nullptr; size_t redo_status_cap = 0;
    // layout of the row counters / status ring as of their last memset (a frame set with the same layout reuses them as they are)
    size_t rows_F = 0; int rows_stride = 0, rows_cap = 0;
    // table path (k_tri_table -> k_pw_rows<TBL>, hg_kernels.h): spans per (frame, triangle, source row), no row lists
    int2 *d_tbl = nullptr; size_t tbl_cap = 0;
    int tbl_stride = 0;                                        // entries per triangle (>= the tallest triangle of the frame set; doubles after an overflow)
    int pw_tri_rows_max = 0;                                   // tallest triangle of the uploaded frames, in rows (host estimate)
    bool pw_table = false;                                     // the current step uses the table path
    bool pw_table_disabled = false;                            // it overflowed twice: row lists from now on
    int pw_table_grown = 0;
    int opt_table = -1;                                        // 1: the table path whenever eligible; -1 / 0: row lists (the default, see run_setup)
    int rows_parity = 0;                                       // which of the two counter sets the current step counts into (ping-pong, hg_kernels.h)
    int opt_tri_threads = -1;                                  // k_tri_spans workgroup size (64 / 128 / 256), -1 by estimate
    int opt_tri_group = -1;                                    // k_tri_spans_grouped: 16 / 64 triangles per workgroup, 0 never, -1 by mesh size
    int opt_rows1_threads = -1;                                // -1 by frame-set size, else 128 or 256
    int opt_col_split = -1;                                    // k_pw_rows workgroups per row group: -1 by frame-set size, else 1, 2 or 4
    // layout estimates of the last frame set, reused for the next set of the same shape (the kernels check the real counts)
    struct LayoutKey { int n = -1, n_tris = -1, max_w = -1, max_h = -1; uint64_t mesh_gen = 0; bool quick = false; } layout_key;
    uint64_t mesh_gen = 0; int layout_age = 0;
    double pw_tri_rows = 0.0; int pw_group_tris = 0;
    long pw_layout_walks = 0;                                  // host walks over the triangles (hg_layout_walks(): tests / bench)

    // geometric frame sets arrive like the piecewise ones: copied into page-locked staging, uploaded stream-ordered, no GPU wait
    // (nothing refers back to a staged geometric set, so a slot is simply reused once its own upload has completed)
    struct GeoStage { uint8_t *h = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; };
    GeoStage geo_stage[8];
    int geo_stage_cur = -1;
    // geometric frames
    int geo_kind = 0;
    bool geo_f32_exact = false;                                // affine matrices hold float values, |x| < 2^28
    std::vector<FrameDesc> geo_frames;
    FrameDesc *d_geo_frames = nullptr; size_t geo_frames_cap = 0;
    double *d_mats = nullptr; size_t mats_cap = 0;
    bool geo_from_points = false;                              // matrices are (re)solved on the device at every warp (hg_geometric_set_frames_points)
    float *d_geo_pts = nullptr; size_t geo_pts_cap = 0;        // F x (from | to) point sets
    int32_t *d_geo_plain = nullptr; size_t geo_plain_cap = 0;  // per-frame "plain division range" flags written by k_solve_frames

    // scratch
    int32_t *d_map32 = nullptr; size_t map32_cap = 0;
    int32_t *d_fmap = nullptr; size_t fmap_cap = 0;            // forward (source-side) triangle map of the current mesh, kept across warps
    bool fmap_valid = false; int fmap_w = 0, fmap_h = 0;
    int32_t *d_win32 = nullptr; size_t win32_cap = 0;
    uint8_t *d_fwd_par = nullptr; size_t fwd_par_cap = 0;      // k_fwd_tiles: FwdParam[n] then FrameDesc[n]
    int32_t *d_fbbox = nullptr; size_t fbbox_cap = 0;          // forward piecewise tiles: per-matrix cell bbox of the forward map (valid with it)
    uint32_t *d_frowoff = nullptr; size_t frowoff_cap = 0;     // ... offset of its per-row extents
    int32_t *d_frowext = nullptr; size_t frowext_cap = 0;      // ... {min mx, max mx} per (matrix index, map row of its bbox)
    bool fwd_rowext_ok = false;
    int32_t *d_ftile_cnt = nullptr; size_t ftile_cnt_cap = 0;  // F x tiles counters (zero between calls)
    int32_t *d_fwd_status = nullptr; size_t fwd_status_cap = 0, fwd_status_stride = 0;   // kFwdStatusRing sets of `stride` status words of tile-binned forward piecewise batches (zero between calls)
    int32_t *d_ftile_ent = nullptr; size_t ftile_ent_cap = 0;  // F x tiles x fwd_pw_cap entries
    double pw_spans_per_window = 0.0;                          // longest row's span count per 256-pixel window (layout heuristic)
    bool pw_quick_layout = false;                              // set around the forward paths' hg_piecewise_set_frames calls
    int fwd_pw_cap = 64;                                       // entries per tile (doubles after an overflow, up to kFwdPwCapMax)
    bool fwd_pw_tiles_disabled = false;                        // overflowed at the largest capacity once: stay with the scatter path for this mesh
    // queued tile-binned forward piecewise batches: status set `slot` of the forward status ring, frame set in staging slot `stage`
    struct FwdPending { uint8_t *out = nullptr; int n = 0; int slot = 0; int stage = -1; int max_src_x = 0, max_src_y = 0; };
    std::vector<FwdPending> fwd_pending;
    int fwd_slot = 0;
    int opt_fwd_tiles = -1;                                    // forward paths: -1 auto, 0 scatter + gather, 1 tiles whenever admissible
    int fwd_last_kernel = 0;                                   // 1 scatter + gather, 2 k_fwd_tiles (hg_last_kernel-style tap for the tests)
    int16_t *d_map16 = nullptr; size_t map16_cap = 0;
    uint8_t *d_out_tmp = nullptr; size_t out_tmp_cap = 0;

    // timing of the dominant kernel: a ring of event pairs recorded around each launch of it
    static constexpr int kEvRing = 256;
    bool timing = false;
    hipEvent_t ev0[kEvRing] = {}, ev1[kEvRing] = {};
    long ev_count = 0;                                         // launches recorded since timing was (re)enabled
};

static int time_begin(hg_ctx *c);
static int time_end(hg_ctx *c);

static int fail(hg_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    g_err = msg;
    return code;
}

#define HIP_TRY(c, expr)                                                                                     \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail((c), HG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
    } while (0)

#define HG_TRY(expr) do { int s_ = (expr); if (s_ != HG_OK) return s_; } while (0)

template <typename T>
static int ensure(hg_ctx *c, T *&p, size_t &cap, size_t need)
{
    if (need <= cap) return HG_OK;
    const size_t n = std::max(need, cap + cap / 2);          // geometric growth from the OLD capacity
    if (p) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(p)); p = nullptr; cap = 0; }
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T));
    if (e != hipSuccess) return fail(c, HG_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    p = static_cast<T *>(q); cap = n;
    return HG_OK;
}

static int bind(hg_ctx *c)
{
    if (!c) return fail(nullptr, HG_ERR_INVALID, "ctx is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    return HG_OK;
}

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
    void *ptrs[] = { c->d_src, c->d_tris, c->d_set, c->d_trir, c->d_segs, c->d_fwd, c->d_inv, c->d_status, c->d_rowcnt, c->d_rowent, c->d_tbl,
                     c->d_geo_frames, c->d_mats, c->d_geo_pts, c->d_geo_plain, c->d_map32, c->d_fmap, c->d_win32, c->d_fwd_par, c->d_fbbox, c->d_frowoff, c->d_frowext, c->d_ftile_cnt, c->d_fwd_status, c->d_ftile_ent, c->d_map16, c->d_out_tmp };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->h_status) (void)hipHostFree(c->h_status);
    for (hg_ctx::Stage &st : c->stage) if (st.h) (void)hipHostFree(st.h);
    for (hg_ctx::GeoStage &gs : c->geo_stage) { if (gs.h) (void)hipHostFree(gs.h); if (gs.done) (void)hipEventDestroy(gs.done); }
    { void *rp[] = { c->d_redo_frame, c->d_redo_dst, c->d_redo_trir, c->d_redo_segs, c->d_redo_fwd, c->d_redo_inv, c->d_redo_status };
      for (void *q : rp) if (q) (void)hipFree(q); }
    for (int i = 0; i < hg_ctx::kEvRing; i++) { if (c->ev0[i]) (void)hipEventDestroy(c->ev0[i]); if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]); }
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
    HIP_TRY(c, hipStreamSynchronize(c->stream));
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
extern "C" int hg_last_forward_kernel(hg_ctx *c) { return c ? c->fwd_last_kernel : 0; }

extern "C" int hg_last_piecewise_table(hg_ctx *c) { return c && c->pw_table ? 1 : 0; }
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
    else if (!std::strcmp(key, "sgpr_cap")) c->opt_sgpr_cap = value;
    else if (!std::strcmp(key, "xcc_rotate")) c->opt_xcc_rotate = value;
    else if (!std::strcmp(key, "table")) { c->opt_table = value; c->pw_table_disabled = false; }
    else if (!std::strcmp(key, "tri_threads")) c->opt_tri_threads = (value == 64 || value == 128 || value == 256) ? value : -1;
    else if (!std::strcmp(key, "tri_group")) c->opt_tri_group = value < 0 ? -1 : (value >= 64 ? 64 : (value ? 16 : 0));
    else if (!std::strcmp(key, "rows1_threads")) c->opt_rows1_threads = (value == 128 || value == 256) ? value : -1;
    else if (!std::strcmp(key, "col_split")) c->opt_col_split = (value == 1 || value == 2 || value == 4) ? value : -1;
    else if (!std::strcmp(key, "lds_pad")) c->opt_lds_pad = std::min(std::max(value, -1), 40);
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

static int time_begin(hg_ctx *c)
{
    if (c->timing) HIP_TRY(c, hipEventRecord(c->ev0[c->ev_count % hg_ctx::kEvRing], c->stream));
    return HG_OK;
}

static int time_end(hg_ctx *c)
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
static int fill_frames(hg_ctx *c, std::vector<FrameDesc> &v, const hg_geom *geoms, const size_t *offs, int n)
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

// ------------------------------------------------------------------------------------------------ affine / projective
static bool geo_plain_division(const double *m, const hg_geom &g) { return projective_plain_range(m, g.x_off, g.y_off, g.obj_w, g.obj_h); }   // hg_math.h

extern "C" int hg_projective_plain_range(const double *m, hg_geom geom)
{
    return (m && geo_plain_division(m, geom)) ? 1 : 0;
}

extern "C" int hg_selftest_division(hg_ctx *c, uint64_t samples, uint64_t seed, uint64_t *mismatches)
{
    HG_TRY(bind(c));
    if (!mismatches) return fail(c, HG_ERR_INVALID, "mismatches is NULL");
    unsigned long long *d = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&d), sizeof(unsigned long long)));
    *mismatches = run_selftest_division(seed, samples, d, c->stream);
    HIP_TRY(c, hipFree(d));
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

// Next staging slot with room for `bytes` (waits only if the upload that last used this slot -- eight sets ago -- is still queued).
static int geo_stage_slot(hg_ctx *c, size_t bytes, hg_ctx::GeoStage **out)
{
    const int slot = (c->geo_stage_cur + 1) % 8;
    hg_ctx::GeoStage &gs = c->geo_stage[slot];
    if (!gs.done) HIP_TRY(c, hipEventCreateWithFlags(&gs.done, hipEventDisableTiming));
    if (gs.used) HIP_TRY(c, hipEventSynchronize(gs.done));
    if (bytes > gs.cap) {
        if (gs.h) { HIP_TRY(c, hipHostFree(gs.h)); gs.h = nullptr; gs.cap = 0; }
        void *q = nullptr;
        hipError_t e = hipHostMalloc(&q, bytes + bytes / 4, hipHostMallocDefault);
        if (e != hipSuccess) return fail(c, HG_ERR_NOMEM, std::string("hipHostMalloc (frame-set staging): ") + hipGetErrorString(e));
        gs.h = static_cast<uint8_t *>(q); gs.cap = bytes + bytes / 4;
    }
    c->geo_stage_cur = slot;
    *out = &gs;
    return HG_OK;
}

extern "C" int hg_geometric_set_frames(hg_ctx *c, int kind, const double *m, const hg_geom *geoms, const size_t *offs, int n)
{
    HG_TRY(bind(c));
    if ((kind != HG_AFFINE && kind != HG_PROJECTIVE) || !m || !geoms || n <= 0) return fail(c, HG_ERR_INVALID, "hg_geometric_set_frames: bad arguments");
    // Transactional: the live frame set is dropped first and the new one only becomes visible once validation, every
    // allocation and the uploads have been queued; after a failure the next *_frames_device call returns HG_ERR_STATE
    // instead of launching F new frames against buffers sized for the old set.  No GPU wait (see hg_piecewise_set_frames).
    c->geo_frames.clear();
    std::vector<FrameDesc> fresh;
    HG_TRY(fill_frames(c, fresh, geoms, offs, n));
    HG_TRY(ensure(c, c->d_geo_frames, c->geo_frames_cap, (size_t)n));
    HG_TRY(ensure(c, c->d_mats, c->mats_cap, (size_t)n * 8));
    const size_t fd_bytes = sizeof(FrameDesc) * (size_t)n, m_bytes = sizeof(double) * 8 * (size_t)n;
    hg_ctx::GeoStage *gs = nullptr;
    HG_TRY(geo_stage_slot(c, fd_bytes + m_bytes, &gs));
    std::memcpy(gs->h, fresh.data(), fd_bytes);
    std::memcpy(gs->h + fd_bytes, m, m_bytes);
    HIP_TRY(c, hipMemcpyAsync(c->d_geo_frames, gs->h, fd_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_mats, gs->h + fd_bytes, m_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(gs->done, c->stream)); gs->used = true;
    c->geo_frames.swap(fresh);
    c->geo_kind = kind; c->geo_from_points = false;
    bool exact = kind == HG_AFFINE;
    for (int f = 0; f < n && exact; f++) {
        for (int k = 0; k < 6; k++) exact = exact && (double)(float)m[8 * f + k] == m[8 * f + k];
        exact = exact && std::abs((int64_t)geoms[f].x_off) + std::max(geoms[f].obj_w, 0) < (1 << 28);
    }
    if (kind == HG_PROJECTIVE) {
        exact = true;
        for (int f = 0; f < n && exact; f++) exact = geo_plain_division(m + 8 * f, geoms[f]);
    }
    c->geo_f32_exact = exact;                               // projective: "every division of the frame set is in the plain range"
    return HG_OK;
}

extern "C" int hg_geometric_set_frames_points(hg_ctx *c, int kind, const float *from, const float *to, const hg_geom *geoms, const size_t *offs, int n)
{
    HG_TRY(bind(c));
    if ((kind != HG_AFFINE && kind != HG_PROJECTIVE) || !from || !to || !geoms || n <= 0) return fail(c, HG_ERR_INVALID, "hg_geometric_set_frames_points: bad arguments");
    const size_t per = kind == HG_AFFINE ? 6 : 8;
    c->geo_frames.clear();                                  // transactional, like hg_geometric_set_frames
    std::vector<FrameDesc> fresh;
    HG_TRY(fill_frames(c, fresh, geoms, offs, n));
    HG_TRY(ensure(c, c->d_geo_frames, c->geo_frames_cap, (size_t)n));
    HG_TRY(ensure(c, c->d_mats, c->mats_cap, (size_t)n * 8));
    HG_TRY(ensure(c, c->d_geo_pts, c->geo_pts_cap, (size_t)n * 16));
    HG_TRY(ensure(c, c->d_geo_plain, c->geo_plain_cap, (size_t)n));
    const size_t fd_bytes = sizeof(FrameDesc) * (size_t)n, p_bytes = sizeof(float) * per * (size_t)n;
    hg_ctx::GeoStage *gs = nullptr;
    HG_TRY(geo_stage_slot(c, fd_bytes + 2 * p_bytes, &gs));
    std::memcpy(gs->h, fresh.data(), fd_bytes);
    std::memcpy(gs->h + fd_bytes, from, p_bytes);
    std::memcpy(gs->h + fd_bytes + p_bytes, to, p_bytes);
    HIP_TRY(c, hipMemcpyAsync(c->d_geo_frames, gs->h, fd_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_geo_pts, gs->h + fd_bytes, p_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_geo_pts + (size_t)n * 8, gs->h + fd_bytes + p_bytes, p_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(gs->done, c->stream)); gs->used = true;
    c->geo_frames.swap(fresh);
    c->geo_kind = kind; c->geo_from_points = true;
    bool exact = kind == HG_AFFINE;                          // affine: the solve stores float32 values; x stays below 2^28?
    for (int f = 0; f < n && exact; f++) exact = std::abs((int64_t)geoms[f].x_off) + std::max(geoms[f].obj_w, 0) < (1 << 28);
    c->geo_f32_exact = exact;
    return HG_OK;
}

extern "C" int hg_get_geometric_matrices(hg_ctx *c, double *out, int n_frames)
{
    HG_TRY(bind(c));
    if (!out || n_frames <= 0 || (size_t)n_frames != c->geo_frames.size()) return fail(c, HG_ERR_INVALID, "hg_get_geometric_matrices: n_frames must equal the uploaded frame count");
    if (c->geo_from_points)
        launch_solve_frames(c->geo_kind, c->d_geo_pts, c->d_geo_pts + c->geo_frames.size() * 8, c->d_geo_frames, c->d_mats, c->d_geo_plain, n_frames, c->stream);
    HIP_TRY(c, hipMemcpyAsync(out, c->d_mats, sizeof(double) * 8 * n_frames, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HG_OK;
}

extern "C" int hg_warp_inverse_geometric_frames_device(hg_ctx *c, void *d_out)
{
    HG_TRY(bind(c));
    if (!d_out) return fail(c, HG_ERR_INVALID, "d_out is NULL");
    if (!c->d_img) return fail(c, HG_ERR_STATE, "no source image: call hg_set_image first");
    if (c->geo_frames.empty()) return fail(c, HG_ERR_STATE, "no frames: call hg_geometric_set_frames first");
    int mw = 0, mh = 0;
    for (const FrameDesc &d : c->geo_frames) { mw = std::max(mw, d.obj_w); mh = std::max(mh, d.obj_h); }
    // the reference re-solves the inverse matrix from the swapped point sets at the head of every warp (:994): so does the step
    if (c->geo_from_points)
        launch_solve_frames(c->geo_kind, c->d_geo_pts, c->d_geo_pts + c->geo_frames.size() * 8, c->d_geo_frames, c->d_mats, c->d_geo_plain,
                            (int)c->geo_frames.size(), c->stream);
    HG_TRY(time_begin(c));
    launch_geo(c->geo_kind, c->geo_f32_exact, c->d_geo_frames, c->d_mats, (int)c->geo_frames.size(), mw, mh, c->d_img, c->W, c->H,
               c->n_imgs, (uint64_t)c->img_stride, static_cast<uint8_t *>(d_out),
               (c->geo_from_points && c->geo_kind == HG_PROJECTIVE) ? c->d_geo_plain : nullptr, c->opt_geo_nw,
               c->n_imgs > 1 ? 0 : c->xcc_log2,              // (one source per frame: plain block order measured faster, 0.234 -> 0.199 ms on C2)
               c->stream);
    HG_TRY(time_end(c));
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

extern "C" int hg_warp_inverse_geometric_batch_device(hg_ctx *c, int kind, const double *m, const hg_geom *geoms,
                                                      const size_t *offs, int n, void *d_out)
{
    HG_TRY(hg_geometric_set_frames(c, kind, m, geoms, offs, n));
    return hg_warp_inverse_geometric_frames_device(c, d_out);
}

extern "C" int hg_warp_inverse_geometric_device(hg_ctx *c, int kind, const double *m, hg_geom geom, void *d_out)
{
    if (!m) return fail(c, HG_ERR_INVALID, "m is NULL");
    double m8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    std::memcpy(m8, m, sizeof(double) * (kind == HG_AFFINE ? 6 : 8));
    const size_t zero = 0;
    return hg_warp_inverse_geometric_batch_device(c, kind, m8, &geom, &zero, 1, d_out);
}

extern "C" int hg_warp_inverse_geometric(hg_ctx *c, int kind, const double *m, hg_geom geom, uint8_t *out_host)
{
    HG_TRY(bind(c));
    if (!out_host) return fail(c, HG_ERR_INVALID, "out is NULL");
    if (geom.obj_w <= 0 || geom.obj_h <= 0) return HG_OK;
    const size_t bytes = (size_t)geom.obj_w * geom.obj_h * 4;
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, bytes));
    HG_TRY(hg_warp_inverse_geometric_device(c, kind, m, geom, c->d_out_tmp));
    HIP_TRY(c, hipMemcpyAsync(out_host, c->d_out_tmp, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HG_OK;
}

// ------------------------------------------------------------------------------------------------ piecewise affine
// NaN is a legal (if useless) coordinate -- the reference then simply draws nothing for that triangle -- but magnitudes beyond
// kMaxCoord (Infinity included) are refused: the row loops of the rasterisers are bounded under that assumption (hg_math.h).
static bool coords_ok(const float *p, size_t n)
{
    for (size_t i = 0; i < n; i++) if (std::fabs((double)p[i]) > kMaxCoord) return false;      // (NaN compares false)
    return true;
}

extern "C" int hg_piecewise_set_mesh(hg_ctx *c, const float *src, int n_pts, const uint32_t *tris, int n_tris, int msx, int msy)
{
    HG_TRY(bind(c));
    if (!src || n_pts <= 0 || n_tris < 0 || (!tris && n_tris > 0)) return fail(c, HG_ERR_INVALID, "hg_piecewise_set_mesh: bad arguments");
    if (!coords_ok(src, (size_t)n_pts * 2))
        return fail(c, HG_ERR_INVALID, "hg_piecewise_set_mesh: a source coordinate is infinite or beyond 2^24 in magnitude");
    // bindings re-send the mesh on every warp (the reference keeps it cached, :742, :758): an identical mesh keeps the
    // device copies and what was derived from them (the forward triangle map)
    if (c->have_mesh && n_pts == c->n_pts && n_tris == c->n_tris && msx == c->min_src_x && msy == c->min_src_y &&
        std::memcmp(src, c->h_src.data(), sizeof(float) * 2 * (size_t)n_pts) == 0 &&
        (n_tris == 0 || std::memcmp(tris, c->h_tris.data(), sizeof(uint32_t) * 3 * (size_t)n_tris) == 0))
        return HG_OK;
    HG_TRY(hg_sync(c));
    HG_TRY(ensure(c, c->d_src, c->src_cap, (size_t)n_pts * 2));
    HG_TRY(ensure(c, c->d_tris, c->tris_cap, (size_t)std::max(n_tris, 1) * 3));
    HIP_TRY(c, hipMemcpyAsync(c->d_src, src, sizeof(float) * 2 * n_pts, hipMemcpyHostToDevice, c->stream));
    if (n_tris > 0) HIP_TRY(c, hipMemcpyAsync(c->d_tris, tris, sizeof(uint32_t) * 3 * n_tris, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->n_pts = n_pts; c->n_tris = n_tris; c->min_src_x = msx; c->min_src_y = msy;
    c->h_tris.assign(tris, tris + (size_t)3 * n_tris);
    c->h_src.assign(src, src + (size_t)2 * n_pts);
    c->fmap_valid = false;
    c->have_mesh = true;
    c->mesh_gen++;
    c->fwd_pw_tiles_disabled = false; c->fwd_pw_cap = 64;     // (learned on the previous mesh)
    c->pw_frames.clear(); c->pw_setup_done = false;
    return HG_OK;
}

// Largest number of triangles whose fillTriangle row range (:1113-1120) covers one output row, over the uploaded frames:
// an estimate of the longest per-row span list, used ONLY to pick k_pw_rows' layout (4 rows per workgroup with 64 LDS
// slots each, or 1 row with all 256); the kernel checks the real counts and is exact either way.
static int max_row_cover(const hg_ctx *c, const float *dst, double *mean_tri_rows, int *max_group_tris, double *mean_shear, int *max_tri_rows, double *fill)
{
    double tallest = 0.0, band_worst = 1.0;
    const int nb = 1 << c->xcc_log2;
    std::vector<double> band((size_t)nb, 0.0);
    int worst = 0, worst_group = 0;
    double rows_total = 0.0, tris_total = 0.0, shear_total = 0.0, shear_n = 0.0;
    std::vector<int> diff, tdiff, starts;
    const int shear_stride = std::max(1, c->n_tris / 1024);
    for (size_t f = 0; f < c->pw_frames.size(); f++) {
        const FrameDesc &fd = c->pw_frames[f];
        if (fd.obj_w <= 0 || fd.obj_h <= 0) continue;
        const float *dp = dst + f * (size_t)c->n_pts * 2;
        diff.assign((size_t)fd.obj_h + 2, 0);
        starts.assign((size_t)fd.obj_h + 2, 0);
        tdiff.assign((size_t)fd.obj_h + 2, 0);
        for (int t = 0; t < c->n_tris; t++) {
            double lo = INFINITY, hi = -INFINITY;
            bool ok = true;
            double sx[3], sy[3], dx[3], dy[3];
            for (int k = 0; k < 3; k++) {
                const uint32_t v = c->h_tris[3 * (size_t)t + k];
                if (v >= (uint32_t)c->n_pts) { ok = false; break; }
                const double y = dp[2 * (size_t)v + 1];
                if (!(y == y)) { ok = false; break; }
                lo = std::min(lo, y); hi = std::max(hi, y);
                sx[k] = c->h_src[2 * (size_t)v]; sy[k] = c->h_src[2 * (size_t)v + 1]; dx[k] = dp[2 * (size_t)v]; dy[k] = y;
            }
            if (!ok) continue;
            if (t % shear_stride == 0) {   // |d(source row) / d(output x)| of the triangle's inverse map (a sample is enough): how many source lines 64 consecutive output
                // pixels spread over (plain doubles: an estimate, never used for pixels)
                const double e1x = sx[1] - sx[0], e1y = sy[1] - sy[0], e2x = sx[2] - sx[0], e2y = sy[2] - sy[0];
                const double f1x = dx[1] - dx[0], f1y = dy[1] - dy[0], f2x = dx[2] - dx[0], f2y = dy[2] - dy[0];
                const double D = e1x * e2y - e2x * e1y;
                const double a = (f1x * e2y - f2x * e1y) / D, cc = (e1x * f2x - e2x * f1x) / D;
                const double b = (f1y * e2y - f2y * e1y) / D, d = (e1x * f2y - e2x * f1y) / D;
                const double sh = std::fabs(b / (a * d - b * cc));
                if (sh == sh && sh < 1e6) { shear_total += sh; shear_n += 1.0; }
            }
            // rows [trunc(minY), ceil(maxY)) - yOff, one more below for spans that spill over the row end (x-offset quirk)
            const double a = std::max(std::trunc(lo) - fd.y_off, 0.0), b = std::min(std::ceil(hi) - fd.y_off + 1.0, (double)fd.obj_h);
            if (!(a < b)) continue;
            diff[(size_t)a] += 1; diff[(size_t)b] -= 1;
            rows_total += b - a; tris_total += 1.0;
            tallest = std::max(tallest, std::min(std::ceil(hi) - std::trunc(lo), 1.0e6));       // rows of fillTriangle's loop :1113-1120

            // tighter, for the triangles-per-group estimate: a triangle has spans on the integer rows inside [minY, maxY]
            // (:1179; triangles that only touch a row at a vertex between two integers do not count), plus the spill row when
            // the window is offset in x
            const double ta = std::max(std::ceil(lo) - fd.y_off, 0.0), tb = std::min(std::floor(hi) - fd.y_off + (fd.x_off != 0 ? 2.0 : 1.0), (double)fd.obj_h);
            if (ta < tb) { tdiff[(size_t)ta] += 1; tdiff[(size_t)tb] -= 1; starts[(size_t)ta] += 1; }
        }
        int run = 0, trun = 0;
        int group = 0;                                       // triangles with spans in the 4-row group the row belongs to
        std::fill(band.begin(), band.end(), 0.0);
        for (int r = 0; r < fd.obj_h; r++) {
            run += diff[r];
            trun += tdiff[r];
            worst = std::max(worst, run);
            band[(size_t)((int64_t)r * nb / fd.obj_h)] += 2.0 + run;          // a row's cost ~ a constant + its spans (layout heuristic only)
            group = (r % kRowGroup == 0) ? trun : group + starts[r];
            worst_group = std::max(worst_group, group);
        }
        double bsum = 0.0, bmax = 0.0;
        for (double v : band) { bsum += v; bmax = std::max(bmax, v); }
        if (bsum > 0) band_worst = std::max(band_worst, bmax * nb / bsum);
    }
    *mean_tri_rows = tris_total > 0 ? rows_total / tris_total : 0.0;
    *max_group_tris = worst_group;
    *mean_shear = shear_n > 0 ? shear_total / shear_n : 0.0;
    *max_tri_rows = (int)tallest;
    *fill = band_worst;                                      // heaviest XCD band / mean band, over the frames (1 = even rows)
    return worst;
}

static int pw_set_frames_impl(hg_ctx *c, const float *dst, const hg_geom *geoms, const size_t *offs, int n);

// Transactional wrapper: if validation, an allocation or an upload fails part-way, the context is left WITHOUT a frame set
// (the next warp returns HG_ERR_STATE) rather than with n new host-side frames over device buffers sized for the old set.
extern "C" int hg_piecewise_set_frames(hg_ctx *c, const float *dst, const hg_geom *geoms, const size_t *offs, int n)
{
    const int rc = pw_set_frames_impl(c, dst, geoms, offs, n);
    if (rc != HG_OK && c) {
        const std::string why = c->err;
        (void)hg_sync(c);                                   // queued runs of the old set are settled against the old set
        c->pw_frames.clear(); c->pw_setup_done = false; c->rows_clean = false;
        c->err = why; g_err = why;
    }
    return rc;
}

static int pw_set_frames_impl(hg_ctx *c, const float *dst, const hg_geom *geoms, const size_t *offs, int n)
{
    HG_TRY(bind(c));
    if (!c->have_mesh) return fail(c, HG_ERR_STATE, "no mesh: call hg_piecewise_set_mesh first");
    if (!dst || !geoms || n <= 0) return fail(c, HG_ERR_INVALID, "hg_piecewise_set_frames: bad arguments");
    if (!coords_ok(dst, (size_t)n * c->n_pts * 2))
        return fail(c, HG_ERR_INVALID, "hg_piecewise_set_frames: a destiny coordinate is infinite or beyond 2^24 in magnitude "
                                       "(the reference's fillTriangle row loop would run for that many rows, forever for Infinity)");
    // Queued runs are NOT waited for: the uploads below are ordered behind them on the stream, and each of them keeps its own
    // staged copy of the set it warped (deferred redo, hg_sync).  Only a staging slot that a queued run still refers to forces a
    // settlement first (a caller that uploads 64 sets per run).
    const size_t T = (size_t)std::max(c->n_tris, 1), F = (size_t)n;
    const int slot = (c->stage_cur + 1) % (int)kStatusRing;
    for (const hg_ctx::Pending &pd : c->pw_pending_out) if (pd.stage == slot) { HG_TRY(hg_sync(c)); break; }
    for (const hg_ctx::FwdPending &pd : c->fwd_pending) if (pd.stage == slot) { HG_TRY(hg_sync(c)); break; }   // (their redo reads the staged set too)
    std::vector<FrameDesc> fresh;
    HG_TRY(fill_frames(c, fresh, geoms, offs, n));
    HG_TRY(ensure(c, c->d_set, c->set_cap, sizeof(FrameDesc) * F + sizeof(float) * 2 * c->n_pts * F));
    HG_TRY(ensure(c, c->d_trir, c->trir_cap, F * T));
    HG_TRY(ensure(c, c->d_segs, c->segs_cap, F * T * 3));
    HG_TRY(ensure(c, c->d_fwd, c->fwd_cap, F * T * 6));
    HG_TRY(ensure(c, c->d_inv, c->inv_cap, F * T * kInvStride));
    HG_TRY(ensure(c, c->d_status, c->status_cap, F));
    if (F * kStatusRing > c->h_status_cap) {
        HG_TRY(hg_sync(c));
        if (c->h_status) HIP_TRY(c, hipHostFree(c->h_status));
        c->h_status = nullptr; c->h_status_cap = 0;
        void *q = nullptr;
        HIP_TRY(c, hipHostMalloc(&q, sizeof(int32_t) * F * kStatusRing, hipHostMallocDefault));
        c->h_status = static_cast<int32_t *>(q); c->h_status_cap = F * kStatusRing;
    }
    hg_ctx::Stage &st = c->stage[slot];
    const size_t fd_bytes = sizeof(FrameDesc) * F, pt_bytes = sizeof(float) * 2 * c->n_pts * F;
    if (fd_bytes + pt_bytes > st.cap) {
        // (an older upload out of this slot may still be queued; no queued run refers to it -- checked above -- but the DMA does)
        if (st.h) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipHostFree(st.h)); st.h = nullptr; st.cap = 0; }
        void *q = nullptr;
        const size_t want = fd_bytes + pt_bytes + (fd_bytes + pt_bytes) / 4;
        hipError_t e = hipHostMalloc(&q, want, hipHostMallocDefault);
        if (e != hipSuccess) return fail(c, HG_ERR_NOMEM, std::string("hipHostMalloc (frame-set staging): ") + hipGetErrorString(e));
        st.h = static_cast<uint8_t *>(q); st.cap = want;
    }
    std::memcpy(st.h, fresh.data(), fd_bytes);
    std::memcpy(st.h + fd_bytes, dst, pt_bytes);
    st.n = n; st.n_pts = c->n_pts;
    static_assert(sizeof(FrameDesc) % 8 == 0, "the destiny points follow the frame records in the same block");
    c->d_pw_frames = reinterpret_cast<FrameDesc *>(c->d_set); c->d_dst = reinterpret_cast<float *>(c->d_set + fd_bytes);
    HIP_TRY(c, hipMemcpyAsync(c->d_set, st.h, fd_bytes + pt_bytes, hipMemcpyHostToDevice, c->stream));    // (one DMA: the staged block has the device layout)
    c->pw_frames.swap(fresh);
    c->stage_cur = slot;
    double tri_rows = 0.0, shear = 0.0;
    int group_tris = 0, max_w = 0, cover = 0, tall = 0;
    double fill = 1.0;
    int64_t total_px = 0;
    for (const FrameDesc &d : c->pw_frames) { max_w = std::max(max_w, d.obj_w); if (d.obj_w > 0 && d.obj_h > 0) total_px += (int64_t)d.obj_w * d.obj_h; }
    int max_h = 0;
    for (const FrameDesc &d : c->pw_frames) if (d.obj_w > 0) max_h = std::max(max_h, d.obj_h);
    const bool quick = c->pw_quick_layout || (total_px < ((int64_t)4 << 20) && (int64_t)F * c->n_tris > 4096);
    // The host walk over every triangle of every frame only picks kernel LAYOUTS (rows per workgroup, entry format, k_pw_patch);
    // the kernels check the real counts and flag what does not fit.  A caller that uploads fresh points for the same mesh and
    // the same window shape every step (the reference's loop, test/benchmark.js:107-110) therefore keeps the previous estimate:
    // same frame count, same mesh, window extents within 1/16; re-walked every 256 sets and whenever a run had to be redone.
    hg_ctx::LayoutKey key;
    key.n = n; key.n_tris = c->n_tris; key.max_w = max_w; key.max_h = max_h; key.mesh_gen = c->mesh_gen; key.quick = quick;
    const hg_ctx::LayoutKey &ok = c->layout_key;
    const bool same_shape = ok.n == key.n && ok.n_tris == key.n_tris && ok.mesh_gen == key.mesh_gen && ok.quick == key.quick &&
                            std::abs(ok.max_w - key.max_w) * 16 <= ok.max_w && std::abs(ok.max_h - key.max_h) * 16 <= ok.max_h &&
                            c->layout_age < 256;
    if (same_shape) {
        cover = c->pw_cover; tri_rows = c->pw_tri_rows; group_tris = c->pw_group_tris; shear = c->pw_shear; tall = c->pw_tri_rows_max; fill = c->pw_fill;
        c->layout_age++;
    } else if (quick) {
        // small frames of a dense mesh (the README's 400x400 / 23 000-triangle benchmark): walking every triangle on the host
        // would cost more than the frame (and the forward paths, which only need the per-triangle solves, skip the walk); guess the row density from the triangle count (the span lists grow if it was low)
        cover = (int)(2.5 * std::sqrt((double)c->n_tris));
        group_tris = 1 << 30;                               // (no k_pw_patch without the real estimate)
        tri_rows = 64.0;
        tall = 0;                                           // (unknown: no table path without the walk)
    } else {
        cover = max_row_cover(c, dst, &tri_rows, &group_tris, &shear, &tall, &fill);
        c->pw_layout_walks++;
    }
    if (!same_shape) { c->layout_key = key; c->layout_age = 0; }
    c->pw_tri_rows = tri_rows; c->pw_group_tris = group_tris; c->pw_tri_rows_max = tall; c->pw_fill = fill;
    c->pw_cover = cover;
    c->pw_spans_per_window = max_w > 0 ? (double)cover * 256.0 / (double)max_w : 0.0;
    c->pw_row_group = cover <= 56 ? kRowGroup : 1;
    {   // few rows in total (a single 4K frame has 560 four-row groups for 256 CUs): one row per workgroup fills the chip better
        int64_t groups = 0;
        for (const FrameDesc &d : c->pw_frames) if (d.obj_w > 0 && d.obj_h > 0) groups += (d.obj_h + kRowGroup - 1) / kRowGroup;
        if (groups < c->opt_min_row_groups) c->pw_row_group = 1;
    }
    // dense rows that still fit the patch kernel's LDS budget, sheared enough for 2-D gather patches to pay.  Measured
    // (k_pw_rows one row per workgroup -> k_pw_patch): C5, shear 0.39, cover 190: 0.63 -> 0.50 ms; 4K 60x60 grid, 0.15, 148:
    // 0.58 -> 0.50; 40x40, 0.16, 98: 0.47 -> 0.45; 32x32, 0.18, 92: 0.43 -> 0.42; but 24x24, 0.08, 60: 0.37 -> 0.38 and C5
    // without its shear, 0.04, 150 (5 spans per 256-pixel window): 0.35 -> 0.37.  Regardless of shear it also wins when many
    // narrow spans share a window (k_pw_rows tests every span of a window on all four pixels of every lane, k_pw_patch only
    // the spans of the lane's 64-pixel bin): lens-distortion style 64x36 grid on 4K, shear 0.02, 8.5 spans per window:
    // 0.73 -> 0.49 ms.  Layout choice only: the kernels check the real counts.
    c->pw_patch = cover > 56 && cover <= kPatchMaxRowSpans && group_tris <= kPatchMaxGroupTris && max_w <= kPatchMaxW &&
                  (int64_t)cover * 64 <= (int64_t)8 * max_w &&          // spans per 64-pixel bin ~ cover * 64 / width: overfull bins are slow
                  (shear >= 0.1 || (int64_t)cover * 256 >= (int64_t)6 * max_w) && !c->pw_patch_disabled;
    // one source per frame (decided at run time, hg_set_images_device may follow): k_pw_patch whenever the frame set fits it, see patch_preferred()
    {
        int64_t groups = 0;
        for (const FrameDesc &d : c->pw_frames) if (d.obj_w > 0 && d.obj_h > 0) groups += (d.obj_h + kRowGroup - 1) / kRowGroup;
        c->pw_patch_fits = cover <= kPatchMaxRowSpans && group_tris <= kPatchMaxGroupTris && max_w <= kPatchMaxW && max_w >= 256 &&
                           (int64_t)cover * 64 <= (int64_t)8 * max_w && groups >= c->opt_min_row_groups && !c->pw_patch_disabled;
    }
    c->pw_shear = shear;
    // beyond that budget, up to ~480 spans per row: the same kernel without matrix records in LDS (pixels read them from global)
    c->pw_patch_dense = !c->pw_patch && cover > kPatchMaxRowSpans && cover <= kPatchMaxRowSpansDense && max_w <= kPatchMaxW &&
                        (int64_t)cover * 64 <= (int64_t)8 * max_w && !c->pw_patch_disabled;
    if (c->pw_patch_dense) c->pw_patch = true;
    // k_tri_spans: one thread per triangle row and round.  Measured round 3 (step ms, 64 / 128 / 256 threads): C5 (~150 rows per
    // triangle, 8 frames) 0.475 / 0.463 / 0.489, C3 (~300 rows, 64 frames) 0.612 / 0.589 / 0.603; a single 4K frame 25.4 / 23.4 / 22.6 us
    c->pw_tri_threads = tri_rows <= 96.0 ? 64 : ((int64_t)F * c->n_tris <= 2048 && tri_rows > 128.0 ? 256 : 128);
    if (c->opt_tri_threads > 0) c->pw_tri_threads = c->opt_tri_threads;
    if (cover > 48 && c->row_cap < kRowSpanCapFast) c->row_cap = kRowSpanCapFast;   // dense rows: size the span lists up front
    if (cover > 200 && c->row_cap < kRowSpanCapDense) c->row_cap = kRowSpanCapDense;
    // (the row counters and the status ring are reused as they are when their layout -- frame count, rows per frame, list
    //  capacity, entry format -- is that of the previous set: run_setup())
    c->pw_setup_done = false;
    return HG_OK;
}

extern "C" int hg_piecewise_prepare(hg_ctx *c, const float *dst, hg_geom geom)
{
    const size_t zero = 0;
    return hg_piecewise_set_frames(c, dst, &geom, &zero, 1);
}

static PwMesh mesh_of(const hg_ctx *c)
{
    PwMesh m;
    m.src_pts = c->d_src; m.tris = c->d_tris; m.n_pts = c->n_pts; m.n_tris = c->n_tris;
    m.min_src_x = c->min_src_x; m.min_src_y = c->min_src_y; m.img = c->d_img; m.W = c->W; m.H = c->H;
    m.n_imgs = c->n_imgs; m.img_stride = c->img_stride;
    return m;
}

static PwFrames frames_of(const hg_ctx *c)
{
    PwFrames f;
    f.frames = c->d_pw_frames; f.dst_pts = c->d_dst; f.trir = c->d_trir; f.segs = c->d_segs; f.fwd = c->d_fwd; f.inv = c->d_inv;
    f.status = c->status_ptr ? c->status_ptr : c->d_status; f.n_frames = (int)c->pw_frames.size();
    int mh = 0;
    for (const FrameDesc &d : c->pw_frames) if (d.obj_w > 0) mh = std::max(mh, d.obj_h);
    f.max_obj_h = mh;
    f.row_group = c->pw_row_group;
    f.tri_threads = c->pw_table ? (c->pw_tri_rows_max <= 64 ? 64 : 128) : c->pw_tri_threads;
    // k_tri_spans_grouped where the per-workgroup solves of k_tri_spans dominate the producer (measured round 3, producer us, 64 frames of
    // 4K unless noted, k_tri_spans -> grouped 16 -> 64): 512 triangles 89 -> 68 -> 68, 3200: 222 -> 151 -> 114, 4608: 342 -> 238 -> 169,
    // C5 (8 frames of 5000) 80 -> 57 -> 57; but C3 (200 triangles of 216 rows) 47.7 -> 51.2 and C4 37.1 -> 38.7: their cost is the
    // slot atomics, not the solves.  64 triangles per workgroup need >= ~1000 workgroups to fill the chip.
    { const int64_t ft = (int64_t)c->pw_frames.size() * c->n_tris;
      f.tri_group = c->opt_tri_group >= 0 ? c->opt_tri_group : ((c->n_tris >= 384 && ft >= 2048) ? (ft >= 65536 ? 64 : 16) : 0); }
    // Windows per phase, measured (C3 / C4, 64 frames): shared (cache-resident) source: 2, or 4 when a window holds
    // several spans (C4's face mesh ~4.5, C3 1.5: the longer span walk then overlaps four windows' gathers); one source per
    // frame (HBM-bound): 4 windows per phase AND fewer, deeper waves -- 12-16 KB of idle LDS per workgroup leave 5 of them on a
    // CU instead of 7 (round 3, same box: C3 0.934 -> 0.910 ms, C4 0.406 -> 0.371), where k_pw_patch does not take the frame set anyway.
    // XCD bands: fixed per XCD when every frame reads the same source and the mesh fills its window (the band's source rows then
    // stay in that XCD's L2 from frame to frame), rotating with the frame otherwise (even load; measured in hg_k_piecewise.hip)
    f.xcc_rotate = c->opt_xcc_rotate >= 0 ? (c->opt_xcc_rotate != 0) : (c->n_imgs > 1 || c->pw_fill > 1.08);
    f.xcc_log2 = c->xcc_log2; f.no_hi_bounds = c->opt_hi_bounds ? 0 : 1;
    f.sgpr_cap = c->opt_sgpr_cap >= 0 ? (c->opt_sgpr_cap != 0) : (c->n_imgs <= 1);
    f.lds_pad_kb = c->opt_lds_pad >= 0 ? c->opt_lds_pad : (c->n_imgs > 1 && c->pw_row_group == kRowGroup ? (c->pw_shear >= 0.1 ? 16 : 12) : 0);
    f.lds_pad_patch_kb = c->opt_lds_pad >= 0 ? c->opt_lds_pad : 0;
    {   // small frame sets: split every row group's windows over 2 or 4 workgroups until the launch has ~4000 of them
        int64_t groups = 0;
        const int rg = c->pw_row_group == kRowGroup ? kRowGroup : 1;
        for (const FrameDesc &d : c->pw_frames) if (d.obj_w > 0 && d.obj_h > 0) groups += (d.obj_h + rg - 1) / rg;
        f.col_split = c->opt_col_split > 0 ? c->opt_col_split : 1;
        f.rows1_threads = c->opt_rows1_threads > 0 ? c->opt_rows1_threads : 256;
        (void)groups;
    }
    f.phase = c->opt_phase > 0 ? c->opt_phase : (c->n_imgs > 1 ? 4 : (c->pw_spans_per_window >= 3.0 ? 4 : 2));
    f.patch_blocks = c->opt_phase > 0 ? c->opt_phase : 8;    // (k_pw_patch: measured best in both source layouts, hg_k_patch.hip)
    return f;
}

static RowLists rows_of(const hg_ctx *c)
{
    RowLists r;
    r.ent = c->d_rowent; r.cap = c->row_cap; r.compact = c->pw_compact ? 1 : 0;
    int mh = 0;
    for (const FrameDesc &d : c->pw_frames) if (d.obj_w > 0) mh = std::max(mh, d.obj_h);
    r.row_stride = std::max(mh, 1);
    const size_t set = c->pw_frames.size() * (size_t)r.row_stride;           // two counter sets, then the status ring
    r.cnt = c->d_rowcnt ? c->d_rowcnt + (size_t)c->rows_parity * set : nullptr;
    r.cnt_clear = c->d_rowcnt ? c->d_rowcnt + (size_t)(1 - c->rows_parity) * set : nullptr;
    return r;
}

// Would the next fused warp of this frame set go through k_pw_patch (the parity tap, which passes a map, never does)?
static bool patch_preferred(const hg_ctx *c, bool *global_records)
{
#ifdef HG_EXPERIMENTS
    static const int env_force = getenv("HG_PATCH") ? atoi(getenv("HG_PATCH")) : -1;    // experiments build only: 0 = never, 1 = whenever allowed by size
#else
    constexpr int env_force = -1;                            // the shipped library reads no environment variable
#endif
    const int force = c->opt_patch >= 0 ? c->opt_patch : env_force;
    int mw = 0;
    for (const FrameDesc &d : c->pw_frames) mw = std::max(mw, d.obj_w);
    if (global_records) *global_records = force == 2 ? true : (force == 1 ? false : c->pw_patch_dense);
    // by estimate (dense, sheared rows), and -- measured round 3 -- whenever every frame streams its own source from HBM and the
    // set fits the kernel: its 16 x 4 gather patches and 8-byte lists beat k_pw_rows there even on sparse meshes (same box, one
    // source per frame: C4 0.371 -> 0.330 ms, C3 step 0.973 -> 0.947)
    return c->pw_fast && mw <= kPatchMaxW && !c->pw_patch_disabled && (force >= 0 ? force >= 1 : (c->pw_patch || (c->n_imgs > 1 && c->pw_patch_fits)));
}

// per-frame solves; status words are reset first.  Fast path: k_tri_spans (solves + per-row span lists);
// general path (more than 32767 triangles, huge sources, negative source minimum): k_tri_setup.
static int run_setup(hg_ctx *c)
{
    const size_t F = c->pw_frames.size();
    int mw = 0;
    for (const FrameDesc &d : c->pw_frames) mw = std::max(mw, d.obj_w);
    c->pw_fast = pw_fast_ok(mesh_of(c), mw);
    if (c->pw_fast) {
        // entry format of the span lists (hg_kernels.h): 8 bytes for dense rows and whenever k_pw_patch will read them
        const bool compact = patch_preferred(c, nullptr) || c->pw_cover > 56;
        if (compact != c->pw_compact) { c->pw_compact = compact; c->rows_clean = false; }
        // Table path (k_tri_table -> k_pw_rows<TBL>), option "table" = 1 only: sparse meshes -- where the row lists would carry
        // 32-byte entries -- whose triangles a workgroup can afford to scan (every 4-row group tests all of them); the tallest
        // triangle sizes the table.  Layout choice only: what does not fit (a taller triangle, more spans per row than a packed
        // block holds) flags its frame.  NOT the default: measured round 3 (same box, 64 frames): the producer gets faster (C3
        // k_tri_spans 42 -> k_tri_table 35 us; without its table stores 22) but the consumer's prologue -- scan, candidate list,
        // two more barriers -- costs more than that (C3 warp kernel 583 -> 604 us, C4 244 -> 249, F = 1 step 23.4 -> 24.9 us).
        const size_t T = (size_t)std::max(c->n_tris, 1);
        c->pw_table = !compact && !c->pw_table_disabled && c->opt_table == 1 && c->row_cap <= kRowSpanCapFast && c->n_tris <= 1024 &&
                      c->pw_tri_rows_max > 0 && c->pw_tri_rows_max <= 8192 && (c->pw_row_group == 1 || c->pw_cover <= 56);
        if (c->pw_table) {
            const int want = ((c->pw_tri_rows_max + 8 + 15) / 16) * 16 << c->pw_table_grown;
            if (want > c->tbl_stride) c->tbl_stride = want;
            if (F * T * (size_t)c->tbl_stride > ((size_t)1 << 28)) c->pw_table = false;          // (2 GiB of table: not this path)
        }
        if (c->pw_table) HG_TRY(ensure(c, c->d_tbl, c->tbl_cap, F * T * (size_t)c->tbl_stride));
        RowLists rl = rows_of(c);
        // Two sets of row counters (ping-pong) + kStatusRing sets of per-frame status words share one allocation.  It is zeroed by a
        // memset only when the layout changes (or after a setup whose warp never ran): the warp kernel of a step zeroes the OTHER
        // counter set -- the one the previous step consumed, the one the next step counts into -- and the next status set.
        const int32_t *before = c->d_rowcnt;
        const size_t ent_bytes = c->pw_table ? 0 : F * (size_t)rl.row_stride * rl.cap * (c->pw_compact ? sizeof(RowEnt8) : sizeof(RowEnt));
        if (2 * F * rl.row_stride + kStatusRing * F > c->rowcnt_cap || ent_bytes > c->rowent_cap) HG_TRY(hg_sync(c));   // (queued runs flag into the old ring)
        HG_TRY(ensure(c, c->d_rowcnt, c->rowcnt_cap, 2 * F * rl.row_stride + kStatusRing * F));
        HG_TRY(ensure(c, c->d_rowent, c->rowent_cap, ent_bytes));
        rl = rows_of(c);
        if (before != c->d_rowcnt || c->rows_F != F || c->rows_stride != rl.row_stride || c->rows_cap != rl.cap) c->rows_clean = false;
        c->rows_F = F; c->rows_stride = rl.row_stride; c->rows_cap = rl.cap;
        if (c->rows_clean) { c->status_slot = (c->status_slot + 1) % (int)kStatusRing; c->rows_parity ^= 1; }
        else {
            HG_TRY(hg_sync(c));                              // (queued runs still own status sets)
            c->status_slot = 0; c->rows_parity = 0;
            HIP_TRY(c, hipMemsetAsync(c->d_rowcnt, 0, sizeof(int32_t) * (2 * F * rl.row_stride + kStatusRing * F), c->stream));
        }
        rl = rows_of(c);                                     // (parity settled)
        c->status_base = c->d_rowcnt + 2 * F * rl.row_stride;
        c->status_ptr = c->status_base + (size_t)c->status_slot * F;
        c->status_next = c->status_base + (size_t)((c->status_slot + 1) % (int)kStatusRing) * F;
        if (c->pw_table) {                                   // (the counters stay untouched: clean for whichever path runs next)
            c->rows_clean = true;
            TriTable tb; tb.ent = c->d_tbl; tb.stride = c->tbl_stride;
            launch_tri_table(mesh_of(c), frames_of(c), tb, c->stream);
        } else {
            c->rows_clean = false;                           // dirty until the warp kernel has consumed them
            launch_tri_spans(mesh_of(c), frames_of(c), rl, c->stream);
        }
    } else {
        HG_TRY(hg_sync(c));                                  // (queued fast-path runs are settled against their own status ring first)
        c->status_ptr = c->d_status;
        HIP_TRY(c, hipMemsetAsync(c->d_status, 0, sizeof(int32_t) * F, c->stream));
        launch_tri_setup(mesh_of(c), frames_of(c), c->stream);
    }
    HIP_TRY(c, hipGetLastError());
    c->pw_setup_done = true;
    return HG_OK;
}

static void run_warp(hg_ctx *c, uint8_t *d_out, int16_t *map_out)
{
    bool global_records = false;
    const bool patch = patch_preferred(c, &global_records) && !map_out && c->pw_compact;
    c->pw_used_patch = patch;
    c->pw_last_kernel = patch ? 3 : (c->pw_fast ? (c->pw_row_group == kRowGroup ? 1 : 2) : 4);
    if (patch)           { launch_pw_patch(mesh_of(c), frames_of(c), rows_of(c), d_out, c->status_next, global_records, c->stream); c->rows_clean = true; }
    else if (c->pw_fast) {
        TriTable tb; tb.ent = c->pw_table ? c->d_tbl : nullptr; tb.stride = c->tbl_stride;
        launch_pw_rows(mesh_of(c), frames_of(c), rows_of(c), tb, d_out, map_out, c->status_next, c->stream); c->rows_clean = true;
    }
    else            launch_pw_fused(mesh_of(c), frames_of(c), d_out, map_out, c->stream);
}

static int check_pw_state(hg_ctx *c)
{
    if (!c->d_img) return fail(c, HG_ERR_STATE, "no source image: call hg_set_image first");
    if (!c->have_mesh) return fail(c, HG_ERR_STATE, "no mesh: call hg_piecewise_set_mesh first");
    if (c->pw_frames.empty()) return fail(c, HG_ERR_STATE, "no frame prepared: call hg_piecewise_prepare / hg_piecewise_set_frames first");
    return HG_OK;
}

// one frame through the materialised map (exact for any input)
static int run_frame_via_map(hg_ctx *c, int f, uint8_t *d_out)
{
    const FrameDesc &fd = c->pw_frames[f];
    const size_t n = (fd.obj_w > 0 && fd.obj_h > 0) ? (size_t)fd.obj_w * fd.obj_h : 0;
    if (n == 0) return HG_OK;
    HG_TRY(ensure(c, c->d_map32, c->map32_cap, n));
    PwMesh mesh = mesh_of(c);
    mesh.img = frame_img(mesh, f); mesh.n_imgs = 1;          // this frame's own source
    launch_map_build(mesh, frames_of(c), f, fd, c->d_map32, c->stream);
    launch_pw_from_map(mesh, frames_of(c), f, fd, c->d_map32, d_out, c->stream);
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

// Deferred redo: frame f of the staged set `stage` (the set a queued run warped; newer sets may have been uploaded since)
// through the materialised map, into `d_out` at the frame's own offset.  Self-contained: the frame's window and points go from
// the staging buffer to a one-frame scratch, k_tri_setup solves it there, then rasteriser + pixel loop.  Mesh and source
// image are those of the context (changing either settles queued runs first).
static int redo_frame_staged(hg_ctx *c, int stage, int f, uint8_t *d_out)
{
    const hg_ctx::Stage &st = c->stage[stage];
    if (stage < 0 || !st.h || f >= st.n || st.n_pts != c->n_pts) return fail(c, HG_ERR_STATE, "deferred redo: the staged frame set is gone");
    const FrameDesc fd = reinterpret_cast<const FrameDesc *>(st.h)[f];
    const size_t n = (fd.obj_w > 0 && fd.obj_h > 0) ? (size_t)fd.obj_w * fd.obj_h : 0;
    if (n == 0) return HG_OK;
    const size_t T = (size_t)std::max(c->n_tris, 1);
    HG_TRY(ensure(c, c->d_redo_frame, c->redo_frame_cap, (size_t)1));
    HG_TRY(ensure(c, c->d_redo_dst, c->redo_dst_cap, (size_t)c->n_pts * 2));
    HG_TRY(ensure(c, c->d_redo_trir, c->redo_trir_cap, T));
    HG_TRY(ensure(c, c->d_redo_segs, c->redo_segs_cap, T * 3));
    HG_TRY(ensure(c, c->d_redo_fwd, c->redo_fwd_cap, T * 6));
    HG_TRY(ensure(c, c->d_redo_inv, c->redo_inv_cap, T * kInvStride));
    HG_TRY(ensure(c, c->d_redo_status, c->redo_status_cap, (size_t)1));
    HG_TRY(ensure(c, c->d_map32, c->map32_cap, n));
    const float *pts = reinterpret_cast<const float *>(st.h + sizeof(FrameDesc) * (size_t)st.n) + (size_t)f * c->n_pts * 2;
    HIP_TRY(c, hipMemcpyAsync(c->d_redo_frame, st.h + sizeof(FrameDesc) * (size_t)f, sizeof(FrameDesc), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_redo_dst, pts, sizeof(float) * 2 * c->n_pts, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_redo_status, 0, sizeof(int32_t), c->stream));
    PwMesh mesh = mesh_of(c);
    mesh.img = frame_img(mesh, f); mesh.n_imgs = 1;          // this frame's own source
    PwFrames fr = frames_of(c);
    fr.frames = c->d_redo_frame; fr.dst_pts = c->d_redo_dst; fr.trir = c->d_redo_trir; fr.segs = c->d_redo_segs; fr.fwd = c->d_redo_fwd;
    fr.inv = c->d_redo_inv; fr.status = c->d_redo_status; fr.n_frames = 1; fr.max_obj_h = fd.obj_h;
    launch_tri_setup(mesh, fr, c->stream);
    launch_map_build(mesh, fr, 0, fd, c->d_map32, c->stream);
    launch_pw_from_map(mesh, fr, 0, fd, c->d_map32, d_out, c->stream);
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

// The forward counterpart: frame f of the staged set through k_fwd_scatter_pw + k_fwd_gather with its own matrices (solved in the
// one-frame scratch), over the context's forward map.
static int redo_forward_frame_staged(hg_ctx *c, int stage, int f, int max_src_x, int max_src_y, uint8_t *d_out)
{
    if (stage < 0 || !c->fmap_valid) return fail(c, HG_ERR_STATE, "deferred forward redo: the staged frame set is gone");
    const hg_ctx::Stage &st = c->stage[stage];
    if (!st.h || f >= st.n || st.n_pts != c->n_pts) return fail(c, HG_ERR_STATE, "deferred forward redo: the staged frame set is gone");
    const FrameDesc fd = reinterpret_cast<const FrameDesc *>(st.h)[f];
    const size_t n = (fd.obj_w > 0 && fd.obj_h > 0) ? (size_t)fd.obj_w * fd.obj_h : 0;
    if (n == 0) return HG_OK;
    const size_t T = (size_t)std::max(c->n_tris, 1);
    HG_TRY(ensure(c, c->d_redo_frame, c->redo_frame_cap, (size_t)1));
    HG_TRY(ensure(c, c->d_redo_dst, c->redo_dst_cap, (size_t)c->n_pts * 2));
    HG_TRY(ensure(c, c->d_redo_trir, c->redo_trir_cap, T));
    HG_TRY(ensure(c, c->d_redo_segs, c->redo_segs_cap, T * 3));
    HG_TRY(ensure(c, c->d_redo_fwd, c->redo_fwd_cap, T * 6));
    HG_TRY(ensure(c, c->d_redo_inv, c->redo_inv_cap, T * kInvStride));
    HG_TRY(ensure(c, c->d_redo_status, c->redo_status_cap, (size_t)1));
    HG_TRY(ensure(c, c->d_win32, c->win32_cap, n));
    const float *pts = reinterpret_cast<const float *>(st.h + sizeof(FrameDesc) * (size_t)st.n) + (size_t)f * c->n_pts * 2;
    HIP_TRY(c, hipMemcpyAsync(c->d_redo_frame, st.h + sizeof(FrameDesc) * (size_t)f, sizeof(FrameDesc), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_redo_dst, pts, sizeof(float) * 2 * c->n_pts, hipMemcpyHostToDevice, c->stream));
    PwFrames fr = frames_of(c);
    fr.frames = c->d_redo_frame; fr.dst_pts = c->d_redo_dst; fr.trir = c->d_redo_trir; fr.segs = c->d_redo_segs; fr.fwd = c->d_redo_fwd;
    fr.inv = c->d_redo_inv; fr.status = c->d_redo_status; fr.n_frames = 1; fr.max_obj_h = fd.obj_h;
    launch_tri_setup(mesh_of(c), fr, c->stream);
    launch_fwd_pw(c->d_fmap, c->d_redo_fwd, c->d_img, c->W, c->H, c->min_src_x, c->min_src_y, max_src_x - c->min_src_x, max_src_y - c->min_src_y,
                  fd, c->d_win32, d_out, c->stream);
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

extern "C" int hg_warp_inverse_piecewise_frames_device(hg_ctx *c, void *d_out)
{
    HG_TRY(bind(c));
    if (!d_out) return fail(c, HG_ERR_INVALID, "d_out is NULL");
    HG_TRY(check_pw_state(c));
    if (c->pw_pending_out.size() >= kStatusRing - 1 || !c->fwd_pending.empty()) HG_TRY(hg_sync(c));
    // The reference recomputes the per-triangle matrices on every setDestinyPoints and the map + inverses on every
    // warp(): both are part of the per-frame step, so both run here every time.
    HG_TRY(run_setup(c));
    HG_TRY(time_begin(c));
    run_warp(c, static_cast<uint8_t *>(d_out), nullptr);
    HG_TRY(time_end(c));
    HIP_TRY(c, hipGetLastError());
    if (c->pw_fast) c->pw_pending_out.push_back({static_cast<uint8_t *>(d_out), c->status_slot, c->stage_cur});
    else {                                                   // general path: one status set, checked right away
        HIP_TRY(c, hipMemcpyAsync(c->h_status, c->status_ptr, sizeof(int32_t) * c->pw_frames.size(), hipMemcpyDeviceToHost, c->stream));
        c->status_base = nullptr;
        c->pw_pending_out.push_back({static_cast<uint8_t *>(d_out), 0, c->stage_cur});
        HG_TRY(hg_sync(c));
    }
    return HG_OK;
}

extern "C" int hg_sync(hg_ctx *c)
{
    HG_TRY(bind(c));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!c->pw_pending_out.empty()) {
        // frames a fused run flagged (irregular, or a row list overflowed) are redone through the materialised map, into the
        // output of the call that flagged them, in call order
        std::vector<hg_ctx::Pending> pending;
        pending.swap(c->pw_pending_out);
        bool redo = false;
        // (all queued runs share one layout of the status ring: a set with another frame count settles them before it runs)
        const int st0 = pending.front().stage;
        const size_t F = (st0 >= 0 && c->stage[st0].h) ? (size_t)c->stage[st0].n : c->pw_frames.size();
        if (c->status_base) HIP_TRY(c, hipMemcpy(c->h_status, c->status_base, sizeof(int32_t) * F * kStatusRing, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < pending.size(); i++) {
            const hg_ctx::Pending &p = pending[i];
            // a later queued run into the SAME output allocation has overwritten this run's frames already (a caller that reuses one
            // buffer step after step): redoing them now would put stale frames over newer ones
            bool superseded = false;
            for (size_t j = i + 1; j < pending.size() && !superseded; j++) superseded = pending[j].out == p.out;
            for (size_t f = 0; f < F; f++)
                if (c->h_status[(size_t)p.slot * F + f] != FRAME_OK) {
                    redo = true; c->pw_redone++;
                    if (!superseded) HG_TRY(redo_frame_staged(c, p.stage, (int)f, p.out));
                }
        }
        if (redo) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (c->pw_used_patch) c->pw_patch_disabled = true;   // (its limits are tighter than k_pw_rows': do not pay the map path again)
            if (c->pw_fast && c->row_cap < kRowSpanCapDense)      // denser mesh than assumed: larger lists next time
                c->row_cap = c->row_cap < kRowSpanCapFast ? kRowSpanCapFast : kRowSpanCapDense;
            c->layout_age = 1 << 30;                             // ... and a fresh layout estimate for the next frame set
            if (c->pw_table) {                                   // the table path flagged: taller triangles than estimated (or denser rows): once more with twice the stride, then row lists
                if (c->pw_table_grown >= 1) c->pw_table_disabled = true; else c->pw_table_grown++;
            }
        }
    }
    if (!c->fwd_pending.empty()) {
        // tile-binned forward piecewise frames the device flagged (a triangle it could not bound, an overfull tile list): redone
        // through scatter + gather into the output of the call that flagged them, from that call's staged frame set (newer sets
        // may have been uploaded since).  The forward map is the context's: a new mesh settles queued runs first.
        std::vector<hg_ctx::FwdPending> pending;
        pending.swap(c->fwd_pending);
        std::vector<int32_t> st(c->fwd_status_cap);
        HIP_TRY(c, hipMemcpy(st.data(), c->d_fwd_status, sizeof(int32_t) * st.size(), hipMemcpyDeviceToHost));
        bool overflow = false, unbounded = false, any = false;
        for (size_t i = 0; i < pending.size(); i++) {
            const hg_ctx::FwdPending &fp = pending[i];
            bool superseded = false;                             // a later queued batch wrote the same output: its frames are the newer ones
            for (size_t j = i + 1; j < pending.size(); j++) if (pending[j].out == fp.out) superseded = true;
            const int32_t *sf = st.data() + (size_t)fp.slot * c->fwd_status_stride;
            for (int f = 0; f < fp.n; f++) {
                if (sf[f] == 0) continue;
                any = true;
                if (sf[f] & FWD_OVERFLOW) overflow = true;
                if (sf[f] & FWD_FALLBACK) unbounded = true;
                if (superseded) continue;
                c->pw_redone++;
                HG_TRY(redo_forward_frame_staged(c, fp.stage, f, fp.max_src_x, fp.max_src_y, fp.out));
            }
        }
        if (any) {
            HIP_TRY(c, hipMemsetAsync(c->d_fwd_status, 0, sizeof(int32_t) * c->fwd_status_cap, c->stream));   // (zero between calls)
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        if (overflow) {
            if (c->fwd_pw_cap < kFwdPwCapMax) c->fwd_pw_cap = std::min(kFwdPwCapMax, c->fwd_pw_cap * 2);
            else c->fwd_pw_tiles_disabled = true;
        }
        if (unbounded) c->fwd_pw_tiles_disabled = true;      // (a degenerate triangle in this mesh: do not pay for both paths again)
    }
    const int d = c->deferred;
    c->deferred = HG_OK;
    return d;
}

extern "C" int hg_warp_inverse_piecewise_batch_device(hg_ctx *c, const float *dst, const hg_geom *geoms, const size_t *offs, int n, void *d_out)
{
    HG_TRY(hg_piecewise_set_frames(c, dst, geoms, offs, n));
    return hg_warp_inverse_piecewise_frames_device(c, d_out);
}

extern "C" int hg_warp_inverse_piecewise_device(hg_ctx *c, void *d_out) { return hg_warp_inverse_piecewise_frames_device(c, d_out); }

static int single_frame_bytes(hg_ctx *c, size_t *bytes)
{
    HG_TRY(check_pw_state(c));
    if (c->pw_frames.size() != 1) return fail(c, HG_ERR_STATE, "this call needs exactly one prepared frame (hg_piecewise_prepare)");
    const FrameDesc &fd = c->pw_frames[0];
    *bytes = (fd.obj_w > 0 && fd.obj_h > 0) ? (size_t)fd.obj_w * fd.obj_h * 4 : 0;
    return HG_OK;
}

extern "C" int hg_warp_inverse_piecewise(hg_ctx *c, uint8_t *out_host)
{
    HG_TRY(bind(c));
    if (!out_host) return fail(c, HG_ERR_INVALID, "out is NULL");
    size_t bytes = 0;
    HG_TRY(single_frame_bytes(c, &bytes));
    if (bytes == 0) return HG_OK;
    const uint64_t keep = c->pw_frames[0].out_off;
    if (keep != 0) return fail(c, HG_ERR_STATE, "prepared frame has a non-zero output offset");
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, bytes));
    HG_TRY(hg_warp_inverse_piecewise_frames_device(c, c->d_out_tmp));
    HG_TRY(hg_sync(c));
    HIP_TRY(c, hipMemcpy(out_host, c->d_out_tmp, bytes, hipMemcpyDeviceToHost));
    return HG_OK;
}

extern "C" int hg_warp_inverse_piecewise_via_map(hg_ctx *c, uint8_t *out_host)
{
    HG_TRY(bind(c));
    if (!out_host) return fail(c, HG_ERR_INVALID, "out is NULL");
    size_t bytes = 0;
    HG_TRY(single_frame_bytes(c, &bytes));
    if (bytes == 0) return HG_OK;
    HG_TRY(hg_sync(c));
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, bytes));
    HG_TRY(run_setup(c));
    HG_TRY(run_frame_via_map(c, 0, c->d_out_tmp));
    HIP_TRY(c, hipMemcpyAsync(out_host, c->d_out_tmp, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HG_OK;
}

extern "C" int hg_get_tri_map(hg_ctx *c, int16_t *out, size_t len)
{
    HG_TRY(bind(c));
    size_t bytes = 0;
    HG_TRY(single_frame_bytes(c, &bytes));
    const size_t n = bytes / 4;
    if (!out || len != n) return fail(c, HG_ERR_INVALID, "hg_get_tri_map: len must be obj_w*obj_h");
    if (n == 0) return HG_OK;
    HG_TRY(hg_sync(c));
    HG_TRY(run_setup(c));
    HG_TRY(ensure(c, c->d_map32, c->map32_cap, n));
    HG_TRY(ensure(c, c->d_map16, c->map16_cap, n));
    launch_map_build(mesh_of(c), frames_of(c), 0, c->pw_frames[0], c->d_map32, c->stream);
    launch_map_to_i16(c->d_map32, c->d_map16, n, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, c->d_map16, n * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HG_OK;
}

extern "C" int hg_get_tri_map_fused(hg_ctx *c, int16_t *out, size_t len)
{
    HG_TRY(bind(c));
    size_t bytes = 0;
    HG_TRY(single_frame_bytes(c, &bytes));
    const size_t n = bytes / 4;
    if (!out || len != n) return fail(c, HG_ERR_INVALID, "hg_get_tri_map_fused: len must be obj_w*obj_h");
    if (n == 0) return HG_OK;
    HG_TRY(hg_sync(c));
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, bytes));
    HG_TRY(ensure(c, c->d_map16, c->map16_cap, n));
    HG_TRY(run_setup(c));
    run_warp(c, c->d_out_tmp, c->d_map16);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->h_status, c->status_ptr, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->h_status[0] != FRAME_OK)
        return fail(c, HG_ERR_STATE, c->h_status[0] & FRAME_IRREGULAR ? "frame is irregular: the fused path defers it to the map path"
                                                                     : "a row overflowed the fused kernel's LDS span list");
    HIP_TRY(c, hipMemcpy(out, c->d_map16, n * sizeof(int16_t), hipMemcpyDeviceToHost));
    return HG_OK;
}

extern "C" int hg_get_matrices(hg_ctx *c, float *fwd, float *inv)
{
    HG_TRY(bind(c));
    size_t bytes = 0;
    HG_TRY(single_frame_bytes(c, &bytes));
    HG_TRY(hg_sync(c));
    if (!c->pw_setup_done) HG_TRY(run_setup(c));
    const size_t T = (size_t)c->n_tris;
    if (T == 0) return HG_OK;
    if (fwd) HIP_TRY(c, hipMemcpyAsync(fwd, c->d_fwd, sizeof(float) * 6 * T, hipMemcpyDeviceToHost, c->stream));
    std::vector<float> tmp;
    if (inv) { tmp.resize(T * kInvStride); HIP_TRY(c, hipMemcpyAsync(tmp.data(), c->d_inv, sizeof(float) * kInvStride * T, hipMemcpyDeviceToHost, c->stream)); }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (inv) for (size_t t = 0; t < T; t++) std::memcpy(inv + 6 * t, tmp.data() + kInvStride * t, sizeof(float) * 6);
    return HG_OK;
}

// ------------------------------------------------------------------------------------------------ forward (scatter) paths
// Limits shared by the forward (scatter) entry points: the raster rank of a source pixel is an int32 (y*W + x resp. the
// cell of the source-bbox map), the scatter kernels put source rows in grid.y, and every window passes the same checks as the
// inverse paths (fill_frames: 2^31 pixels, offsets within 2^26, 4-byte aligned output offsets).
static int forward_limits(hg_ctx *c, int64_t w, int64_t h, const char *what)
{
    if (w <= 0 || h <= 0) return HG_OK;
    if (w * h >= ((int64_t)1 << 31)) return fail(c, HG_ERR_INVALID, std::string(what) + " has 2^31 pixels or more: the forward path ranks source pixels in 32 bits");
    if (h > 65535) return fail(c, HG_ERR_INVALID, std::string(what) + " is taller than 65535 rows: not supported by the forward path");
    return HG_OK;
}

// Is k_fwd_tiles admissible for this frame (see the kernel: every bound below keeps the rounding error of its candidate
// constraints under the 1/64-pixel widening, and no source pixel further than kFwdWrap columns outside the window)?
// Fills P (matrix, inverse = adjugate / det of the 3x3 form, use_inv only when the inverse reproduces the source corners).
static bool fwd_tile_param(int kind, const double *m, int W, int H, const FrameDesc &fd, FwdParam &P)
{
    if (fd.obj_w < 2 * kFwdWrap || fd.obj_h <= 0 || W <= 0 || H <= 0 || W > 65535 || H > 65535) return false;
    const bool proj = kind == HG_PROJECTIVE;
    for (int k = 0; k < (proj ? 8 : 6); k++) if (!std::isfinite(m[k])) return false;
    const double lim = proj ? 1.0e4 : 1.0e6, flim = proj ? 1.0e5 : 1.0e7;
    for (int k = 0; k < 6; k++) if (std::fabs(m[k]) > lim) return false;
    if (proj && (std::fabs(m[6]) > 0.1 || std::fabs(m[7]) > 0.1)) return false;
    double Hm[9];
    if (proj) { for (int k = 0; k < 8; k++) Hm[k] = m[k]; Hm[8] = 1.0; }
    else { Hm[0] = m[0]; Hm[1] = m[2]; Hm[2] = m[4]; Hm[3] = m[1]; Hm[4] = m[3]; Hm[5] = m[5]; Hm[6] = 0.0; Hm[7] = 0.0; Hm[8] = 1.0; }
    double umin = INFINITY, umax = -INFINITY, cfx[4], cfy[4];
    for (int c = 0; c < 4; c++) {
        const double x = (c & 1) ? W - 1 : 0, y = (c & 2) ? H - 1 : 0;
        const double den = Hm[6] * x + Hm[7] * y + Hm[8];
        if (!(den >= 1.0e-2)) return false;                  // (linear: positive at the corners = positive on the whole image)
        const double fx = (Hm[0] * x + Hm[1] * y + Hm[2]) / den, fy = (Hm[3] * x + Hm[4] * y + Hm[5]) / den;
        if (!(std::fabs(fx) < flim && std::fabs(fy) < flim)) return false;
        cfx[c] = fx; cfy[c] = fy;
        umin = std::min(umin, fx - fd.x_off); umax = std::max(umax, fx - fd.x_off);
    }
    // the image of the source rectangle is the convex hull of its corner images: no rounded x further out than kFwdWrap - 2
    if (umin < -(double)(kFwdWrap - 2) || umax > (double)fd.obj_w + (kFwdWrap - 2)) return false;
    for (int k = 0; k < 8; k++) P.m[k] = k < (proj ? 8 : 6) ? m[k] : 0.0;
    P.use_inv = 0; P.pad = 0;
    const double det = Hm[0] * (Hm[4] * Hm[8] - Hm[5] * Hm[7]) - Hm[1] * (Hm[3] * Hm[8] - Hm[5] * Hm[6]) + Hm[2] * (Hm[3] * Hm[7] - Hm[4] * Hm[6]);
    const double adj[9] = { Hm[4] * Hm[8] - Hm[5] * Hm[7], Hm[2] * Hm[7] - Hm[1] * Hm[8], Hm[1] * Hm[5] - Hm[2] * Hm[4],
                            Hm[5] * Hm[6] - Hm[3] * Hm[8], Hm[0] * Hm[8] - Hm[2] * Hm[6], Hm[2] * Hm[3] - Hm[0] * Hm[5],
                            Hm[3] * Hm[7] - Hm[4] * Hm[6], Hm[1] * Hm[6] - Hm[0] * Hm[7], Hm[0] * Hm[4] - Hm[1] * Hm[3] };
    for (int k = 0; k < 9; k++) P.inv[k] = 0.0;
    if (std::isfinite(det) && std::fabs(det) > 1.0e-12) {
        bool ok = true;
        for (int k = 0; k < 9; k++) { P.inv[k] = adj[k] / det; if (!std::isfinite(P.inv[k])) ok = false; }
        for (int c = 0; c < 4 && ok; c++) {                  // the inverse has to bring the corner images back (1e-3 px), with w > 0
            const double x = (c & 1) ? W - 1 : 0, y = (c & 2) ? H - 1 : 0;
            const double X = P.inv[0] * cfx[c] + P.inv[1] * cfy[c] + P.inv[2], Y = P.inv[3] * cfx[c] + P.inv[4] * cfy[c] + P.inv[5];
            const double Wd = P.inv[6] * cfx[c] + P.inv[7] * cfy[c] + P.inv[8];
            if (!(Wd > 0.0) || !(std::fabs(X / Wd - x) < 1.0e-3) || !(std::fabs(Y / Wd - y) < 1.0e-3)) ok = false;
        }
        P.use_inv = ok ? 1 : 0;
    }
    return true;
}

extern "C" int hg_forward_tiles_admissible(int kind, const double *m, int W, int H, hg_geom geom)
{
    if ((kind != HG_AFFINE && kind != HG_PROJECTIVE) || !m) return 0;
    FrameDesc fd; fd.x_off = geom.x_off; fd.y_off = geom.y_off; fd.obj_w = geom.obj_w; fd.obj_h = geom.obj_h; fd.out_off = 0; fd.map_off = 0;
    FwdParam P;
    if (!fwd_tile_param(kind, m, W, H, fd, P)) return 0;
    return P.use_inv ? 2 : 1;
}

extern "C" int hg_warp_forward_geometric_batch_device(hg_ctx *c, int kind, const double *m, const hg_geom *geoms, const size_t *offs, int n, void *d_out)
{
    HG_TRY(bind(c));
    if ((kind != HG_AFFINE && kind != HG_PROJECTIVE) || !m || !geoms || n <= 0 || !d_out) return fail(c, HG_ERR_INVALID, "hg_warp_forward_geometric: bad arguments");
    if (!c->d_img) return fail(c, HG_ERR_STATE, "no source image: call hg_set_image first");
    if (c->n_imgs > 1) return fail(c, HG_ERR_STATE, "the forward warps take ONE source image (hg_set_images_device with n_images > 1 serves the inverse warps only)");
    HG_TRY(forward_limits(c, c->W, c->H, "the source image"));
    std::vector<FrameDesc> fds;
    HG_TRY(fill_frames(c, fds, geoms, offs, n));
    size_t max_px = 0;
    for (const FrameDesc &fd : fds) if (fd.obj_w > 0 && fd.obj_h > 0) max_px = std::max(max_px, (size_t)fd.obj_w * fd.obj_h);
    if (max_px == 0) return HG_OK;
    HG_TRY(hg_sync(c));
    // Tile-binned gather (k_fwd_tiles, all frames in one launch) when every frame is admissible and the windows are large
    // enough to fill the chip with tiles of a bounded candidate count; otherwise scatter + gather frame after frame.
    if (c->opt_fwd_tiles != 0) {
        std::vector<FwdParam> par((size_t)n);
        bool all = true;
        int64_t tiles = 0;
        int mw = 0, mh = 0;
        for (int f = 0; f < n && all; f++) {
            if (fds[f].obj_w <= 0 || fds[f].obj_h <= 0) { std::memset(&par[f], 0, sizeof(FwdParam)); continue; }
            all = fwd_tile_param(kind, m + 8 * (size_t)f, c->W, c->H, fds[f], par[f]);
            const int64_t t = (int64_t)((fds[f].obj_w + kFwdTileW - 1) / kFwdTileW) * ((fds[f].obj_h + kFwdTileH - 1) / kFwdTileH);
            if (c->opt_fwd_tiles < 0 && (int64_t)c->W * c->H > t * 32768) all = false;   // (few tiles for many source pixels: long serial candidate loops)
            tiles += t; mw = std::max(mw, fds[f].obj_w); mh = std::max(mh, fds[f].obj_h);
        }
        if (all && c->opt_fwd_tiles < 0 && tiles < 160) all = false;     // (measured break-even against the three small scatter-path kernels)
        if (all) {
            FwdBatch batch;
            std::memset(&batch, 0, sizeof batch);
            if (n == 1) { batch.p0 = par[0]; batch.f0 = fds[0]; }
            else {
                const size_t bytes = sizeof(FwdParam) * n + sizeof(FrameDesc) * n;
                HG_TRY(ensure(c, c->d_fwd_par, c->fwd_par_cap, bytes));
                std::vector<uint8_t> blob(bytes);
                std::memcpy(blob.data(), par.data(), sizeof(FwdParam) * n);
                std::memcpy(blob.data() + sizeof(FwdParam) * n, fds.data(), sizeof(FrameDesc) * n);
                HIP_TRY(c, hipMemcpyAsync(c->d_fwd_par, blob.data(), bytes, hipMemcpyHostToDevice, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream)); // caller / local memory is not retained
                batch.params = reinterpret_cast<const FwdParam *>(c->d_fwd_par);
                batch.frames = reinterpret_cast<const FrameDesc *>(c->d_fwd_par + sizeof(FwdParam) * n);
            }
            launch_fwd_tiles(kind, batch, n, mw, mh, c->d_img, c->W, c->H, static_cast<uint8_t *>(d_out), c->stream);
            HIP_TRY(c, hipGetLastError());
            c->fwd_last_kernel = 2;
            return HG_OK;
        }
    }
    c->fwd_last_kernel = 1;
    HG_TRY(ensure(c, c->d_mats, c->mats_cap, (size_t)8 * n));
    HG_TRY(ensure(c, c->d_win32, c->win32_cap, max_px));
    HIP_TRY(c, hipMemcpyAsync(c->d_mats, m, sizeof(double) * 8 * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));         // caller memory is not retained
    c->geo_frames.clear();                               // the uploaded geometric frame set was overwritten
    for (int f = 0; f < n; f++)                          // frames run back to back on the stream (one winner buffer, reused in order)
        launch_fwd_geo(kind, c->d_mats + 8 * (size_t)f, c->d_img, c->W, c->H, fds[f], c->d_win32, static_cast<uint8_t *>(d_out), c->stream);
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

extern "C" int hg_warp_forward_geometric_device(hg_ctx *c, int kind, const double *m, hg_geom geom, void *d_out)
{
    if (!m) return fail(c, HG_ERR_INVALID, "m is NULL");
    double m8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    std::memcpy(m8, m, sizeof(double) * (kind == HG_AFFINE ? 6 : 8));
    const size_t zero = 0;
    return hg_warp_forward_geometric_batch_device(c, kind, m8, &geom, &zero, 1, d_out);
}

extern "C" int hg_warp_forward_geometric(hg_ctx *c, int kind, const double *m, hg_geom geom, uint8_t *out_host)
{
    HG_TRY(bind(c));
    if (!out_host) return fail(c, HG_ERR_INVALID, "hg_warp_forward_geometric: bad arguments");
    if (geom.obj_w <= 0 || geom.obj_h <= 0) return HG_OK;
    const size_t n = (size_t)geom.obj_w * geom.obj_h;
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, n * 4));
    HG_TRY(hg_warp_forward_geometric_device(c, kind, m, geom, c->d_out_tmp));
    HIP_TRY(c, hipMemcpyAsync(out_host, c->d_out_tmp, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HG_OK;
}

// Per-row extents of every matrix index inside its bbox (k_fwd_pw_tiles' candidate bound), built once per forward map when the
// tile path first wants them: offsets are a prefix sum of the bbox heights (host).
static int ensure_fwd_rowext(hg_ctx *c, int map_w, int map_h)
{
    if (c->fwd_rowext_ok || c->n_tris <= 0) return HG_OK;
    std::vector<int32_t> bb((size_t)4 * c->n_tris);
    HIP_TRY(c, hipMemcpyAsync(bb.data(), c->d_fbbox, sizeof(int32_t) * bb.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<uint32_t> off((size_t)c->n_tris);
    uint64_t total = 0;
    for (int t = 0; t < c->n_tris; t++) {
        off[t] = (uint32_t)total;
        if (bb[4 * (size_t)t + 3] >= bb[4 * (size_t)t + 1]) total += (uint64_t)(bb[4 * (size_t)t + 3] - bb[4 * (size_t)t + 1] + 1);
        if (total > ((uint64_t)1 << 27)) { c->fwd_pw_tiles_disabled = true; return HG_OK; }   // fill() wrap-around quirks made the boxes absurdly tall: scatter path
    }
    HG_TRY(ensure(c, c->d_frowoff, c->frowoff_cap, (size_t)c->n_tris));
    HG_TRY(ensure(c, c->d_frowext, c->frowext_cap, (size_t)2 * std::max<uint64_t>(total, 1)));
    HIP_TRY(c, hipMemcpyAsync(c->d_frowoff, off.data(), sizeof(uint32_t) * off.size(), hipMemcpyHostToDevice, c->stream));
    launch_fmap_rowext(c->d_fmap, map_w, map_h, c->d_fbbox, c->d_frowoff, c->d_frowext, (size_t)total, c->n_tris, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));             // (`off` is local)
    c->fwd_rowext_ok = true;
    return HG_OK;
}

// _piecewiseAffineWarp :948-972 for n destination point sets on the current mesh (the caller loop `setDestinyPoints(d_f); warp()`
// when warp() takes the forward path, :421), asynchronous, frames in GPU memory.
extern "C" int hg_warp_forward_piecewise_batch_device(hg_ctx *c, const float *dst_points, int max_src_x, int max_src_y, const hg_geom *geoms,
                                                      const size_t *offs, int n, void *d_out)
{
    HG_TRY(bind(c));
    if (!dst_points || !geoms || n <= 0 || !d_out) return fail(c, HG_ERR_INVALID, "hg_warp_forward_piecewise: bad arguments");
    if (!c->d_img) return fail(c, HG_ERR_STATE, "no source image: call hg_set_image first");
    if (c->n_imgs > 1) return fail(c, HG_ERR_STATE, "the forward warps take ONE source image (hg_set_images_device with n_images > 1 serves the inverse warps only)");
    if (!c->have_mesh) return fail(c, HG_ERR_STATE, "no mesh: call hg_piecewise_set_mesh first");
    const int64_t map_w = (int64_t)max_src_x - c->min_src_x, map_h = (int64_t)max_src_y - c->min_src_y;
    HG_TRY(forward_limits(c, map_w, map_h, "the source-point bounding box"));
    const size_t n_map = (map_w > 0 && map_h > 0) ? (size_t)map_w * map_h : 0;
    // (A) forward triangle map over the source bbox: _buildTrianglesCorrespondencesMatrix :817-832 == the same
    //     rasteriser on the SOURCE triangles with width maxSrcX-minSrcX and y offset minSrcY.  It depends on the mesh only,
    //     so it is kept until the mesh (or the bbox) changes -- like the reference's cached _trianglesCorrespondencesMatrix.
    hg_geom gmap = { 0, c->min_src_y, (int32_t)map_w, (int32_t)map_h };
    const size_t zero = 0;
    if (!c->pw_pending_out.empty()) HG_TRY(hg_sync(c));       // (queued inverse runs: one order of deferred redos per output buffer)
    if (n_map && !(c->fmap_valid && c->fmap_w == map_w && c->fmap_h == map_h)) {
        if (!c->fwd_pending.empty()) HG_TRY(hg_sync(c));     // (queued batches are redone over the map they ran on)
        HG_TRY(hg_piecewise_set_frames(c, c->h_src.data(), &gmap, &zero, 1));
        c->status_ptr = c->d_status;
        HIP_TRY(c, hipMemsetAsync(c->d_status, 0, sizeof(int32_t), c->stream));
        launch_tri_setup(mesh_of(c), frames_of(c), c->stream);
        HG_TRY(ensure(c, c->d_fmap, c->fmap_cap, n_map));
        launch_map_build(mesh_of(c), frames_of(c), 0, c->pw_frames[0], c->d_fmap, c->stream);
        HG_TRY(ensure(c, c->d_fbbox, c->fbbox_cap, (size_t)4 * std::max(c->n_tris, 1)));
        launch_fmap_bbox(c->d_fmap, (int)map_w, (int)map_h, c->d_fbbox, c->n_tris, c->stream);     // (for the tile-binned frames below)
        HIP_TRY(c, hipGetLastError());
        c->fmap_valid = true; c->fmap_w = (int)map_w; c->fmap_h = (int)map_h;
        c->fwd_rowext_ok = false;
    }
    // (B) forward matrices of every frame (:785-804) in one launch, then scatter + gather frame after frame
    c->pw_quick_layout = true;                               // (the inverse kernels' layout estimate is not needed here: no host walk over the triangles)
    const int rc_frames = hg_piecewise_set_frames(c, dst_points, geoms, offs, n);
    c->pw_quick_layout = false;
    HG_TRY(rc_frames);
    c->status_ptr = c->d_status;                             // (k_tri_setup only ORs flags into these words and nothing on the forward path reads them: not cleared)
    launch_tri_setup(mesh_of(c), frames_of(c), c->stream);
    c->pw_setup_done = false;
    size_t max_px = 0;
    for (const FrameDesc &fd : c->pw_frames) if (fd.obj_w > 0 && fd.obj_h > 0) max_px = std::max(max_px, (size_t)fd.obj_w * fd.obj_h);
    if (max_px) {
        // Tile-binned gather (k_fwd_pw_bins + k_fwd_pw_tiles, all frames in two launches) when the batch has enough tiles;
        // frames the device cannot bound (flagged in their status word) are redone through scatter + gather by hg_sync.
        int mw = 0, mh = 0;
        int64_t tiles = 0;
        for (const FrameDesc &fd : c->pw_frames) if (fd.obj_w > 0 && fd.obj_h > 0) {
            mw = std::max(mw, fd.obj_w); mh = std::max(mh, fd.obj_h);
            tiles += (int64_t)((fd.obj_w + kFwdTileW - 1) / kFwdTileW) * ((fd.obj_h + kFwdTileH - 1) / kFwdTileH);
        }
        const int tsx = (mw + kFwdTileW - 1) / kFwdTileW, tsy = (mh + kFwdTileH - 1) / kFwdTileH;
        bool use_tiles = n_map > 0 && c->n_tris > 0 && !c->fwd_pw_tiles_disabled && tsy <= 65535 && map_w <= 65535 &&   // (winner keys: 16 bits per map coordinate)
                         (c->opt_fwd_tiles > 0 || (c->opt_fwd_tiles < 0 && tiles >= 160 && (int64_t)n_map <= tiles * 32768 &&
                                                   (int64_t)c->n_tris * n <= 8 * tiles));   // (measured up to 6 triangles per tile, 4K 96 x 54 cells: 171 -> 92 us per frame; denser: not measured, scatter path)
        if (use_tiles) { HG_TRY(ensure_fwd_rowext(c, (int)map_w, (int)map_h)); use_tiles = c->fwd_rowext_ok; }
        if (use_tiles) {
            // The tile counters and the per-frame status words are ZERO between calls: k_fwd_pw_tiles clears the counter of every
            // tile it consumes, hg_sync the status words it found set -- no memset in front of every batch (two stream operations
            // less per call: they were a fifth of a single 4K frame's time).  Status words: a ring of sets, one per queued batch.
            const size_t per = (size_t)n * tsx * tsy;
            { const size_t cap0 = c->ftile_cnt_cap;
              HG_TRY(ensure(c, c->d_ftile_cnt, c->ftile_cnt_cap, per));
              if (c->ftile_cnt_cap != cap0) HIP_TRY(c, hipMemsetAsync(c->d_ftile_cnt, 0, sizeof(int32_t) * c->ftile_cnt_cap, c->stream)); }
            if (c->fwd_pending.size() >= kFwdStatusRing - 1 || (size_t)n > c->fwd_status_stride) {
                HG_TRY(hg_sync(c));                              // ring full, or a larger batch than the ring's sets were laid out for
                if ((size_t)n > c->fwd_status_stride) {
                    HG_TRY(ensure(c, c->d_fwd_status, c->fwd_status_cap, (size_t)n * kFwdStatusRing));
                    c->fwd_status_stride = c->fwd_status_cap / kFwdStatusRing;
                    HIP_TRY(c, hipMemsetAsync(c->d_fwd_status, 0, sizeof(int32_t) * c->fwd_status_cap, c->stream));
                }
            }
            c->fwd_slot = (c->fwd_slot + 1) % (int)kFwdStatusRing;
            HG_TRY(ensure(c, c->d_ftile_ent, c->ftile_ent_cap, per * (size_t)c->fwd_pw_cap));
            FwdPwTiles p;
            p.fmap = c->d_fmap; p.fwd = c->d_fwd; p.bbox = c->d_fbbox; p.frames = c->d_pw_frames; p.rowext = c->d_frowext; p.rowoff = c->d_frowoff;
            p.tile_cnt = c->d_ftile_cnt; p.tile_ent = c->d_ftile_ent; p.status = c->d_fwd_status + (size_t)c->fwd_slot * c->fwd_status_stride;
            p.T = c->n_tris; p.min_src_x = c->min_src_x; p.min_src_y = c->min_src_y; p.map_w = (int)map_w; p.map_h = (int)map_h;
            p.tsx = tsx; p.tsy = tsy; p.cap = c->fwd_pw_cap;
            launch_fwd_pw_tiles(p, n, mw, mh, c->d_img, c->W, c->H, static_cast<uint8_t *>(d_out), c->stream);
            HIP_TRY(c, hipGetLastError());
            { hg_ctx::FwdPending fp;
              fp.out = static_cast<uint8_t *>(d_out); fp.n = n; fp.slot = c->fwd_slot; fp.stage = c->stage_cur; fp.max_src_x = max_src_x; fp.max_src_y = max_src_y;
              c->fwd_pending.push_back(fp); }
            c->fwd_last_kernel = 2;
            return HG_OK;
        }
        c->fwd_last_kernel = 1;
        HG_TRY(ensure(c, c->d_win32, c->win32_cap, max_px));
        for (int f = 0; f < n; f++)
            launch_fwd_pw(c->d_fmap, c->d_fwd + (size_t)f * c->n_tris * 6, c->d_img, c->W, c->H, c->min_src_x, c->min_src_y, (int)map_w, (int)map_h,
                          c->pw_frames[f], c->d_win32, static_cast<uint8_t *>(d_out), c->stream);
    }
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

extern "C" int hg_warp_forward_piecewise_device(hg_ctx *c, const float *dst_points, int max_src_x, int max_src_y, hg_geom geom, void *d_out)
{
    const size_t zero = 0;
    return hg_warp_forward_piecewise_batch_device(c, dst_points, max_src_x, max_src_y, &geom, &zero, 1, d_out);
}

extern "C" int hg_warp_forward_piecewise(hg_ctx *c, const float *dst_points, int max_src_x, int max_src_y, hg_geom geom, uint8_t *out_host)
{
    HG_TRY(bind(c));
    if (!out_host) return fail(c, HG_ERR_INVALID, "hg_warp_forward_piecewise: bad arguments");
    if (geom.obj_w <= 0 || geom.obj_h <= 0) return HG_OK;
    const size_t n = (size_t)geom.obj_w * geom.obj_h;
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, n * 4));
    HG_TRY(hg_warp_forward_piecewise_device(c, dst_points, max_src_x, max_src_y, geom, c->d_out_tmp));
    HG_TRY(hg_sync(c));                                      // (settles a frame the tile kernels handed to the scatter path)
    HIP_TRY(c, hipMemcpyAsync(out_host, c->d_out_tmp, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HG_OK;
}
