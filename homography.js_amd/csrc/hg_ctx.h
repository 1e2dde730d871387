// hg_ctx.h -- the context behind the C ABI of include/hgwarp.h and the helpers its translation units share
// (hg_api.hip: library / context / buffers / host-side solves / source image; hg_api_geometric.hip; hg_api_piecewise.hip;
// hg_api_forward.hip; hg_api_state.hip).  Internal: nothing here is exported.
#pragma once
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
extern thread_local std::string g_err;   // (hg_api.hip) message of the last failure on this thread, for calls without a context

struct hg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t copy_stream = nullptr;                         // hg_upload_on_copy_stream: uploads that overlap the warp stream's work
    hipEvent_t copy_event = nullptr;                           // hg_fence_copies
    hipStream_t down_stream = nullptr;                         // hg_download_behind_warps: D2H copies that overlap the warp stream's later work
    hipEvent_t down_event = nullptr;
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
    int2 *d_trix = nullptr; size_t trix_cap = 0;
    Seg *d_segs = nullptr; size_t segs_cap = 0;
    float *d_fwd = nullptr; size_t fwd_cap = 0;
    float *d_inv = nullptr; size_t inv_cap = 0;
    int32_t *d_status = nullptr; size_t status_cap = 0;
    int32_t *d_two_round = nullptr; size_t two_round_cap = 0; int32_t pw_gen = 0;   // PwFrames::two_round / gen
    int32_t *status_ptr = nullptr;                             // where this frame set's status words live (d_status, or the tail of d_rowcnt)
    int32_t *h_status = nullptr; size_t h_status_cap = 0;      // pinned
    int32_t *h_flag = nullptr;                                 // pinned, device-visible: set to 1 by any fused kernel that flags a frame (PwFrames::host_flag)
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
    int opt_sub_bands = -1;                                    // option "sub_bands": sub-bands per XCD of the warp kernels on a shared source with fixed bands (0 / 1 off, -1 by the source's size)
    int opt_xcc_rotate = -1;                                   // -1 by estimate, 0 / 1
    int opt_compact = -1;                                      // span-list entry format: 1 = 8-byte entries, 0 = 32-byte entries with the matrix, -1 by estimate
    double pw_shear = 0.0;                                     // mean |d(source row) / d(output x)| of the uploaded frames (layout heuristic)
    bool pw_patch_dense = false;                               // ... only in its global-record variant (up to 511 spans per row)
    bool pw_patch_disabled = false;                            // a group exceeded k_pw_patch's limits once: stay with k_pw_rows for this mesh
    bool pw_used_patch = false;                                // the last fused run went through k_pw_patch
    int pw_last_kernel = 0;                                    // hg_last_piecewise_kernel()
    int32_t pw_last_flag = 0;                                  // status word of the last frame a fused run flagged (bits 4..: which limit, see k_pw_patch<SELF>)
    long pw_redone = 0;                                        // frames redone through the materialised map (hg_redone_frames())
    int opt_min_row_groups = 1152, opt_patch = -1, opt_phase = -1, opt_geo_nw = 8;   // hg_set_option()
    int xcc_log2 = 3;                                          // log2(XCCs of the device): hipDeviceAttributeNumberOfXccs at hg_create, option "xcc"
    int opt_hi_bounds = 1;                                     // 0: fp64 bounds compares instead of the high-dword form (hg_dev.h)
    // fused runs whose per-frame status words have not been checked yet: up to kStatusRing - 1 calls are queued back to back
    // with nothing but their two kernels in the stream; each flags into its own set of status words, read back by hg_sync
    // stage: which staged frame set (points + windows) the run warped; extent / layout: the bytes it writes from `out` on and a hash of
    // its frames' (offset, size) list -- a later call into the SAME layout supersedes its deferred redos frame by frame, any other
    // overlapping writer settles it first (settle_output_conflicts)
    // path: which layout the run took (bit 0 k_pw_patch, bit 1 self-span prologue, bit 2 k_pw_tile): what hg_sync disables when the run exceeded a limit
    struct Pending { uint8_t *out; int slot; int stage; size_t extent; uint64_t layout; uint8_t path; };
    std::vector<Pending> pw_pending_out;
    // Frame sets arrive through a ring of page-locked staging buffers (FrameDesc[F], then the F x n_pts x 2 destination
    // points): hg_piecewise_set_frames copies the caller's arrays there and queues stream-ordered uploads -- it neither waits
    // for the GPU nor keeps caller memory.  A staged set stays intact until every run that used it has been settled, so frames a
    // fused run flagged can still be redone (through the materialised map) after newer sets were uploaded.
    // `done` is recorded behind the slot's upload: the slot's bytes are not rewritten before the DMA that reads them has run
    // (paths that keep no pending record -- the forward scatter path -- could otherwise lap the ring with uploads still queued)
    struct Stage { uint8_t *h = nullptr; size_t cap = 0; int n = 0, n_pts = 0; hipEvent_t done = nullptr; bool used = false; };
    Stage stage[kStatusRing];
    int stage_cur = -1;
    // scratch of the deferred redo (one frame): its FrameDesc, points, solves
    FrameDesc *d_redo_frame = nullptr; size_t redo_frame_cap = 0;
    float *d_redo_dst = nullptr; size_t redo_dst_cap = 0;
    TriRange *d_redo_trir = nullptr; size_t redo_trir_cap = 0;
    int2 *d_redo_trix = nullptr; size_t redo_trix_cap = 0;
    Seg *d_redo_segs = nullptr; size_t redo_segs_cap = 0;
    float *d_redo_fwd = nullptr; size_t redo_fwd_cap = 0;
    float *d_redo_inv = nullptr; size_t redo_inv_cap = 0;
    int32_t *d_redo_status = nullptr; size_t redo_status_cap = 0;
    // reference-state warps (hg_api_state.hip): the map's own point set and triangles, the cached matrices handed over by the caller
    float *d_st_pts = nullptr; size_t st_pts_cap = 0;
    uint32_t *d_st_tris = nullptr; size_t st_tris_cap = 0;
    float *d_st_mats = nullptr; size_t st_mats_cap = 0;
    // layout of the row counters / status ring as of their last memset (a frame set with the same layout reuses them as they are)
    size_t rows_F = 0; int rows_stride = 0, rows_cap = 0;
    // self-span path (k_tri_setup -> k_pw_rows<SELF>, hg_kernels.h): the row workgroups evaluate their own spans, no row lists
    int pw_tri_rows_max = 0;                                   // tallest triangle of the uploaded frames, in rows (host estimate)
    int64_t pw_groups = 0;                                     // 4-row groups of the frame set (layout estimate)
    bool pw_small_set = false;                                 // fewer 4-row groups than "min_row_groups": one row per workgroup, short-latency prologues
    bool pw_self = false;                                      // the current step uses the self-span path
    bool pw_self_disabled = false;                             // a run on it flagged a frame (more candidates / spans than its LDS blocks hold): row lists for this mesh
    bool pw_self_patch = false;                                // ... through k_pw_patch (dense / sheared meshes, one source per frame)
    bool pw_tile = false;                                      // ... through k_pw_tile (8-row x <= 2048-column tiles whose gathers follow the source rows)
    bool pw_tile_disabled = false;                             // a tile exceeded its limits once: k_pw_patch for this mesh
    int pw_last_variant = 0;                                   // variant code of the last piecewise warp kernel launched (launch_pw_rows; hg_last_piecewise_variant)
    int opt_tile = -1;                                         // option "tile": 1 whenever k_pw_patch<SELF> would run, 0 never, -1 by policy
    bool pw_bands = false;                                     // ... with candidate bands (meshes too large for every workgroup to scan)
    int4 *d_bands = nullptr; size_t bands_cap = 0;             // F x n_bands x band_cap entries (hg_kernels.h)
    int band_cap = 0, n_bands = 0;
    int opt_self = -1;                                         // option "self_spans": 1 whenever eligible, 0 never, -1 by policy (run_setup)
    int rows_parity = 0;                                       // which of the two counter sets the current step counts into (ping-pong, hg_kernels.h)
    int opt_tri_group = -1;                                    // k_tri_spans_grouped: 16 / 64 triangles per workgroup, 0 never, -1 by mesh size
    int opt_upload_kernel = -1;                                // frame-set blocks up to 1 MB go up by k_upload (default) instead of hipMemcpyAsync (0)
    int opt_safe_spans = -1;                                   // option "safe_spans": span flags + bounds-test-free windows in k_pw_rows: 1 / 0, -1 by the spans-per-window estimate
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
    struct FwdPending { uint8_t *out = nullptr; int n = 0; int slot = 0; int stage = -1; int max_src_x = 0, max_src_y = 0; size_t extent = 0; uint64_t layout = 0; };
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

inline int fail(hg_ctx *c, int code, const std::string &msg)
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

// A staged block (page-locked, device-visible) to the device, stream-ordered: up to 1 MB by k_upload -- a kernel that reads the host block --,
// beyond that (or with option "upload_kernel" = 0) by the copy engine, whose start-up latency is what a 36-KB frame set paid for (R4.13).
inline int upload_staged(hg_ctx *c, void *d0, const void *s0, size_t b0, void *d1 = nullptr, const void *s1 = nullptr, size_t b1 = 0,
                         void *d2 = nullptr, const void *s2 = nullptr, size_t b2 = 0)
{
    if (b0 + b1 + b2 == 0) return HG_OK;
    if (std::max(b0, std::max(b1, b2)) <= ((size_t)1 << 20) && ((b0 | b1 | b2) & 7) == 0 && c->opt_upload_kernel != 0) {
        UploadSegs sg{{d0, d1, d2}, {s0, s1, s2}, {b0 / 8, b1 / 8, b2 / 8}};
        launch_upload(sg, c->stream);
        HIP_TRY(c, hipGetLastError());
    } else {
        if (b0) HIP_TRY(c, hipMemcpyAsync(d0, s0, b0, hipMemcpyHostToDevice, c->stream));
        if (b1) HIP_TRY(c, hipMemcpyAsync(d1, s1, b1, hipMemcpyHostToDevice, c->stream));
        if (b2) HIP_TRY(c, hipMemcpyAsync(d2, s2, b2, hipMemcpyHostToDevice, c->stream));
    }
    return HG_OK;
}

template <typename T>
inline int ensure(hg_ctx *c, T *&p, size_t &cap, size_t need)
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

inline int bind(hg_ctx *c)
{
    if (!c) return fail(nullptr, HG_ERR_INVALID, "ctx is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    return HG_OK;
}

// ------------------------------------------------------------------------------------------------ shared between the translation units
int time_begin(hg_ctx *c);                                   // hg_api.hip: event pair around the dominant kernel (hg_set_timing)
int time_end(hg_ctx *c);
int fill_frames(hg_ctx *c, std::vector<FrameDesc> &v, const hg_geom *geoms, const size_t *offs, int n);      // hg_api.hip
// The bytes a frame list writes from its output pointer on, and a hash of its (offset, size) pairs (0 is never returned).
void output_layout(const std::vector<FrameDesc> &frames, size_t *extent, uint64_t *layout);                  // hg_api.hip
// Before a call writes [out, out + extent): queued runs whose deferred redo could land on those bytes later are settled now, unless
// the new call has the same base and layout (then hg_sync skips the older run's redo frame by frame: `superseded`).  layout = 0:
// a writer that keeps no pending record (geometric warps, the scatter paths) -- any overlap settles.
int settle_output_conflicts(hg_ctx *c, const void *out, size_t extent, uint64_t layout);                     // hg_api.hip
PwMesh mesh_of(const hg_ctx *c);                             // hg_api_piecewise.hip: kernel argument blocks of the current mesh / frame set
PwFrames frames_of(const hg_ctx *c);
int redo_forward_frame_staged(hg_ctx *c, int stage, int f, int max_src_x, int max_src_y, uint8_t *d_out);    // hg_api_piecewise.hip, beside its inverse twin
