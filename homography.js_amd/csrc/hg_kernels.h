// hg_kernels.h -- device-side data layout and kernel launchers (implemented in hg_k_*.hip, one translation unit per kernel family).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_math.h"

namespace hg {

// One output frame (one destination point set / one matrix): the reference's
// (_xOutputOffset, _yOutputOffset, _objectiveWidth, _objectiveHeight) plus where its RGBA goes.
struct FrameDesc {
    int32_t x_off, y_off, obj_w, obj_h;
    uint64_t out_off;        // byte offset of this frame's RGBA8 in the output allocation
    uint64_t map_off;        // cell offset of this frame in the optional int16 debug-map allocation
};

// Per (frame, triangle) raster parameters produced by k_tri_setup.
//   rows y in [y_min, y_end) are the rows fillTriangle visits (:1113-1120);
//   a, b bound which output rows a row-y span can touch:  (y - yOff) in [r - a, r - b]  (see DESIGN.md §4.2).
struct TriRange { int32_t y_min, y_end, a, b; };

// Per-frame status word written by the kernels.
enum : int32_t {
    FRAME_OK = 0,
    FRAME_IRREGULAR = 1,     // a triangle with non-finite / absurd vertices or wider than the whole map: fused path skipped
    FRAME_LDS_OVERFLOW = 2   // some row crossed more spans than the fused kernel's LDS list holds: frame must be redone
};

constexpr int kRowSpanCap = 1024;   // spans per output row held in LDS by the general fused kernel (k_pw_fused)
constexpr int kRowSpanCapFast = 256; // ... by the fast kernel (k_pw_rows), which also keeps a 48-byte matrix per span
constexpr int kRowSpanCapDense = 512; // ... by its dense instantiation (README-scale meshes: ~23 000 triangles, 200-500 spans per row)
constexpr int kRowGroup = 4;            // output rows per k_pw_rows workgroup (64 LDS slots each in packed mode)
constexpr int kInvStride = 8;       // floats per inverse matrix on the device (6 used; 32-byte rows)

struct PwMesh {                     // source side of the mesh + source image (shared by all frames)
    const float *src_pts;           // n_pts x 2
    const uint32_t *tris;           // n_tris x 3
    int32_t n_pts, n_tris;
    int32_t min_src_x, min_src_y;
    const uint8_t *img;             // RGBA8 source, W*H*4 bytes (image 0)
    int32_t W, H;
    int32_t n_imgs;                 // >= 1: frame f reads image f % n_imgs (1 = one source shared by every frame)
    uint64_t img_stride;            // bytes between consecutive source images (hg_set_images_device)
};

// source image of frame f
__host__ __device__ __forceinline__ const uint8_t *frame_img(const PwMesh &m, int f)
{
    return m.n_imgs > 1 ? m.img + (uint64_t)(f % m.n_imgs) * m.img_stride : m.img;
}

struct PwFrames {                   // per-frame device arrays, frame-major
    const FrameDesc *frames;
    const float *dst_pts;           // F x n_pts x 2
    TriRange *trir;                 // F x n_tris
    int2 *trix;                     // F x n_tris: {floor(min x) - 1, ceil(max x) + 1} of the triangle's destiny vertices (column reach of its spans; k_pw_tile)
    Seg *segs;                      // F x n_tris x 3
    float *fwd;                     // F x n_tris x 6
    float *inv;                     // F x n_tris x kInvStride
    int32_t *status;                // F
    int32_t *two_round;             // F: == gen when k_tri_setup met a triangle (with rows) of the frame whose :1383 sums are not exact (affine_fusable, hg_math.h);
    int32_t gen;                    //    any other value: the row kernel may use one fma per coordinate for the whole frame.  gen: this step's number (never reset: no clearing pass)
    int32_t *host_flag;             // page-locked, device-visible word set to 1 whenever a kernel flags a frame (nullptr: none): lets hg_sync skip reading the status ring
    int32_t n_frames;
    int32_t max_obj_h;              // max over frames (grid size)
    int32_t row_group;              // output rows per k_pw_rows workgroup: kRowGroup (sparse rows) or 1 (dense meshes)
    int32_t tri_threads;            // k_tri_spans workgroup size: 128, or 64 when the triangles are short (one row per thread)
    int32_t tri_group;              // 16 / 64: k_tri_spans_grouped (that many triangles per workgroup, their solves one per lane) instead of k_tri_spans; 0: k_tri_spans
    int32_t phase;                  // k_pw_rows: windows whose gathers are issued before their stores (1, 2 or 4)
    int32_t xcc_rotate;             // 1: XCD x takes band (x + frame) mod XCCs instead of band x (uneven rows, or no source shared between frames)
    int32_t sub_bands;              // host: sub-bands per XCD (0 / 1: none); the launcher derives sub_groups from it for its kernel's row-group height
    int32_t sub_groups;             // shared source, fixed bands: 0, or the row groups of a SUB-band.  The frame's rows are cut into sub-bands of that many groups, dealt to the XCDs round
                                    // robin (sub-band j -> XCD j mod XCCs), and an XCD takes ALL FRAMES of one of its sub-bands before the next (loop interchange): the slice of the source a
                                    // sub-band reads (a few hundred rows) then stays in that XCD's 4 MiB L2 from frame to frame, which a whole band of a 4K source does not (R6.12)
    int32_t xcc_log2;               // log2 of the device's XCC count (8 on an unpartitioned MI355X): block id -> XCD row band
    int32_t lds_pad_kb;             // KB of unused dynamic LDS per k_pw_rows workgroup (caps the workgroups resident per CU: row lists with one source per frame)
    int32_t sgpr_cap;               // k_pw_rows PH = 2: the 80-SGPR instantiation (8 workgroups per CU); host: shared source only
    int32_t no_hi_bounds;           // option "hi_bounds" = 0: keep the fp64 bounds compares (parity suite runs both forms)
    int32_t safe_spans;             // 1: the row kernel flags every span whose two end pixels pass the source bounds test and runs windows made of such spans
                                    // without the per-pixel test (k_pw_rows); 0: no flags (rows of many short spans, where flagging costs more than it saves)
    int32_t safe_spans_patch;       // the same flags in k_pw_patch<SELF> (per 64-pixel x 4-row block)
    int32_t self_spans;             // 1: no row lists -- k_tri_setup ran, the
                                    // warp kernel's workgroups evaluate the spans of their own rows in their prologue
    // Candidate bands of the self-span path for meshes too large to scan per workgroup (k_tri_setup files every triangle under the
    // bands of 1 << band_rows_log2 output rows it can reach; a row workgroup tests only its band's entries).  band_ent == nullptr:
    // workgroups scan trir[frame][0 .. n_tris).  Counters: the row-counter arrays (unused without row lists), index frame * band_stride + band.
    int4 *band_ent;                 // F x n_bands x band_cap entries of TWO int4: {triangle, y_min, y_end, a & 0xffff | b << 16}, {xlo, xhi, 0, 0}
    int32_t *band_cnt;
    int32_t band_stride, n_bands, band_cap, band_rows_log2;
};

// Per-output-row span lists of the fast path (k_tri_spans -> k_pw_rows), in global memory.
// Two entry formats, chosen per frame set (RowLists::compact):
//   full    32 bytes: cells [lo,hi) of the row (16 bits each), triangle id, the triangle's inverse matrix.  Sparse rows
//           (<= 56 spans, k_pw_rows' packed 4-row groups): the matrix arrives with the entry, no dependent load in the prologue.
//   compact  8 bytes: cells + id only; consumers fetch the matrix by id from the per-(frame, triangle) tap array fr.inv (L2
//           resident).  Dense rows (k_pw_patch, k_pw_rows one row per workgroup): a quarter of the list traffic -- C5's
//           k_pw_patch 0.508 -> 0.415 ms, k_tri_spans 79 -> 68 us -- at the price of one dependent L2 load per row, which costs
//           sparse C3 3 % and is why both formats exist.
struct RowEnt { uint32_t lo_hi; int32_t id; float m[6]; };
struct RowEnt8 { uint32_t lo_hi; int32_t id; };
static_assert(sizeof(RowEnt) == 32 && sizeof(RowEnt8) == 8, "span entry sizes");
struct RowLists {
    int32_t *cnt;            // F x row_stride: this step's counters (k_tri_spans counts into them, the warp kernel reads them)
    int32_t *cnt_clear;      // the other of the two counter sets, same layout: the warp kernel zeroes it for the NEXT step (ping-pong)
    void *ent;               // F x row_stride x cap entries of 32 (RowEnt) or 8 (RowEnt8) bytes
    int32_t compact;         // 1: RowEnt8
    int32_t row_stride;      // >= max obj_h
    int32_t cap;             // entries per row
};

// k_tri_setup: per (frame, triangle): forward affine (:785-804, :1265-1306), its inverse (:1036-1038, :1345-1365),
// edge equations (:1141-1151) and row range (:1113-1115).
void launch_tri_setup(const PwMesh &mesh, const PwFrames &fr, hipStream_t stream);
struct UploadSegs { void *dst[3]; const void *src[3]; size_t n8[3]; };     // (dst, page-locked device-visible src, count of 8-byte words)
void launch_upload(const UploadSegs &sg, hipStream_t stream);              // small host -> device copies by ONE kernel launch

// k_pw_fused: _inversePiecewiseAffineWarp :1029-1058 for all frames, one workgroup per output row, without a
// materialised triangle map.  map_out (optional, int16 per output pixel) receives the per-pixel triangle id the
// lookup resolved == the reference's _trianglesCorrespondencesMatrix.
void launch_pw_fused(const PwMesh &mesh, const PwFrames &fr, uint8_t *out, int16_t *map_out, hipStream_t stream);

// Fast path (see hg_k_piecewise.hip): eligibility, span-list build (includes the per-triangle solves), row warp.
bool pw_fast_ok(const PwMesh &mesh, int max_obj_w);
void launch_tri_spans(const PwMesh &mesh, const PwFrames &fr, const RowLists &rl, hipStream_t stream);
int launch_pw_rows(const PwMesh &mesh, const PwFrames &fr, const RowLists &rl, uint8_t *out, int16_t *map_out, int32_t *status_next, hipStream_t stream);
// Dense meshes (64..199 spans per row, obj_w <= 8192): 4 rows per workgroup, 16 x 4 pixel gather patches, one matrix record
// per triangle of the group.  Same row lists, same status protocol as launch_pw_rows; no map tap.
// (limits on the HOST ESTIMATES, which run ~10 % above the real counts the kernel enforces: 199 spans per row, 208 triangles
// per group; a frame set that does exceed those is redone once through the map path and the context drops back to k_pw_rows)
constexpr int kPatchMaxW = 8192, kPatchMaxRowSpans = 215, kPatchMaxGroupTris = 225;
// global_records: the variant for up to 511 spans per row whose pixels read their matrix from the tap array instead of LDS
constexpr int kPatchMaxRowSpansDense = 480;
int launch_pw_patch(const PwMesh &mesh, const PwFrames &fr, const RowLists &rl, uint8_t *out, int32_t *status_next, bool global_records, hipStream_t stream);
// Sheared meshes (self-span path only): tiles of 8 rows x <= 2048 columns whose gathers follow the source rows (hg_k_tile.hip).
constexpr int kTileRowSpanCap = 96;       // k_pw_tile: spans per row and tile (its LDS span blocks); beyond: the frame is flagged and redone, the mesh goes back to k_pw_patch
int launch_pw_tile(const PwMesh &mesh, const PwFrames &fr, const RowLists &rl, uint8_t *out, int max_obj_w, int32_t *status_next, hipStream_t stream);

// Materialised-map path for ONE frame (index f): map32 := -1; atomicMax rasteriser (:845-861 + :1111-1126); then
// the pixel loop :1042-1056 reading the map.
void launch_map_build(const PwMesh &mesh, const PwFrames &fr, int f, const FrameDesc &fd, int32_t *map32, hipStream_t stream);
void launch_pw_from_map(const PwMesh &mesh, const PwFrames &fr, int f, const FrameDesc &fd, const int32_t *map32,
                        uint8_t *out, hipStream_t stream);
void launch_map_to_i16(const int32_t *map32, int16_t *map16, size_t n, hipStream_t stream);
void launch_map_max_i16(const int32_t *map32, size_t n, int32_t *out, hipStream_t stream);      // *out = largest (int16) value of the first n cells, -1 if none
void launch_fill_i32(int32_t *p, size_t n, int32_t v, hipStream_t stream);      // grid-stride fill (map / winner-buffer initialisation)

// k_geo: _inverseGeometricWarp pixel loop :997-1011 for all frames.  mats = F x 8 doubles (inverse matrices).
// f32_exact: every affine matrix entry is a float value and |x| < 2^28 (lets the kernel use an exact-product fma).
// n_imgs / img_stride: frame f reads the source at img + (f % n_imgs) * img_stride.
void launch_geo(int kind, bool f32_exact, const FrameDesc *frames, const double *mats, int n_frames, int max_w, int max_h,
                const uint8_t *img, int W, int H, int n_imgs, uint64_t img_stride, uint8_t *out, const int32_t *plain, int nw, int xcc_log2, bool rotate_bands, hipStream_t stream);
// Per-frame matrix solves on the device (one lane per frame): kind 1 = projective 8x8 DLT in numeric.js' LU order, 4 points
// per set; kind 0 = affine closed form, 3 points per set.  mats = F x 8 doubles; plain[f] = 1 where the projective frame's
// window stays in the plain division range (launch_geo then takes the per-frame flag instead of a host proof).
void launch_solve_frames(int kind, const float *from, const float *to, const FrameDesc *frames, double *mats, int32_t *plain, int n, hipStream_t stream);
// (f32_exact doubles as "plain division range proved" for kind 1: see geo_plain_division() in hg_api_geometric.hip)
// div2_plain vs IEEE division on `samples` pseudo-random operand triples; returns the number of mismatching quotients
unsigned long long run_selftest_division(uint64_t seed, uint64_t samples, unsigned long long *d_counter, hipStream_t stream);

// Forward (scatter) paths, SURVEY.md §8f-1: winner buffer `win` = obj_w*obj_h int32 scratch.
// Tile-binned forward warp (k_fwd_tiles): per frame the forward matrix (8 doubles, affine in m[0..5]), the adjugate of its 3x3
// form (any positive multiple of the inverse; use_inv = 0 when it is not trustworthy: tiles then scan every source row).
struct FwdParam { double m[8]; double inv[9]; int32_t use_inv, pad; };
constexpr int kFwdTileW = 64, kFwdTileH = 64, kFwdWrap = 32;
// params == nullptr: one frame, carried by value (p0, f0) in the kernel arguments
struct FwdBatch { const FwdParam *params; const FrameDesc *frames; FwdParam p0; FrameDesc f0; };
// n_imgs / img_stride (all forward launchers that take them): frame f reads the source at img + (f % n_imgs) * img_stride
void launch_fwd_tiles(int kind, const FwdBatch &batch, int n_frames, int max_w, int max_h,
                      const uint8_t *img, int n_imgs, uint64_t img_stride, int W, int H, uint8_t *out, hipStream_t stream);
// Tile-binned forward PIECEWISE warp: k_fmap_bbox (once per mesh: bounding box, in map cells, of the cells the forward
// triangle map assigns to each matrix index), k_fwd_pw_bins (per frame and triangle: which output tiles its pixels can reach,
// per aliasing shift k of the flat index; frames it cannot bound are flagged for the scatter path), k_fwd_pw_tiles (per tile:
// gather the candidates of the filed (triangle, k) entries, winners in LDS).
enum : int32_t { FWD_FALLBACK = 1, FWD_OVERFLOW = 2 };
constexpr int kFwdPwCapMax = 256;
struct FwdPwTiles {
    const int32_t *fmap; const float *fwd; const int32_t *bbox; const FrameDesc *frames;
    const int32_t *rowext;      // per (matrix index t, row of its bbox): {min mx, max mx} of its cells in that map row, at rowoff[t] + (my - bbox.cy0)
    const uint32_t *rowoff;
    int32_t *tile_cnt, *tile_ent, *status;
    int32_t *host_flag;         // page-locked, device-visible word set to 1 by any kernel that flags a frame (nullptr: none); see PwFrames::host_flag
    int32_t T, min_src_x, min_src_y, map_w, map_h, tsx, tsy, cap;
};
void launch_fmap_bbox(const int32_t *fmap, int map_w, int map_h, int32_t *bbox, int T, hipStream_t stream);
void launch_fmap_rowext(const int32_t *fmap, int map_w, int map_h, const int32_t *bbox, const uint32_t *rowoff, int32_t *rowext, size_t total_rows, int T, hipStream_t stream);
void launch_fwd_pw_tiles(const FwdPwTiles &p, int n_frames, int max_w, int max_h, const uint8_t *img, int n_imgs, uint64_t img_stride, int W, int H, uint8_t *out, hipStream_t stream);
void launch_fwd_geo(int kind, const double *d_mat, const uint8_t *img, int W, int H, const FrameDesc &fd, int32_t *win, uint8_t *out, hipStream_t stream);
void launch_fwd_pw(const int32_t *fmap, const float *fwd, const uint8_t *img, int W, int H, int min_src_x, int min_src_y, int map_w, int map_h,
                   const FrameDesc &fd, int32_t *win, uint8_t *out, hipStream_t stream);

} // namespace hg
