/*
 * hgwarp.h -- C ABI of libhgwarp.so, the MI355X (gfx950) inverse-warp engine that sits behind the Homography.js
 * `Homography` class for its per-pixel hot path.
 *
 * The reference (Eric-Canas/Homography.js v1.8.0) has no FFI seam: the seam is the ES-module class itself.  Each
 * entry point below therefore names the reference *function* it replaces (file:line into the reference's
 * Homography.js).  The bindings that call this ABI are
 *     homography.js_amd/csrc/hgwarp_napi.c   (N-API addon used by the drop-in JS class homography.js_amd/js/Homography.mjs)
 *     homography.js_amd/hgwarp.py            (ctypes; used by tests/, bench.py and __graft_entry__.py)
 * and INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns an int status (HG_OK == 0); hg_last_error() gives the text of the last failure;
 *   - plain pointers + sizes, no C++/torch types; caller memory is never retained after return unless the function
 *     name ends in `_device` and says "aliased";
 *   - one hg_ctx per Homography instance; a ctx is bound to one GPU and one HIP stream and is not thread-safe;
 *     distinct ctxs are independent;
 *   - images are RGBA8, row-major, 4*W*H bytes (Uint8ClampedArray of an ImageData, reference :297-299);
 *   - points are interleaved x,y float32 (the reference's Float32Array, :220/:339); triangles are 3 uint32 vertex ids
 *     each (Delaunator's `.triangles`, :1217); per-triangle matrices are 6 float32 [a,b,c,d,e,f] with
 *     x' = a*x + c*y + e, y' = b*x + d*y + f (:1297-1304);
 *   - functions ending in `_device` take/leave data in GPU memory and are asynchronous on the ctx stream
 *     (hg_sync() waits and reports deferred errors); the others are synchronous like the reference's warp();
 *   - there is NO CPU fallback: without a usable gfx950 device hg_create() fails and nothing warps.
 */
#ifndef HGWARP_H
#define HGWARP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HG_VERSION 100          /* 0.1.0 */

enum {
    HG_OK = 0,
    HG_ERR_INVALID = 1,         /* bad argument (NULL pointer, negative size, unknown kind) */
    HG_ERR_HIP = 2,             /* a HIP runtime call failed (text in hg_last_error) */
    HG_ERR_NO_DEVICE = 3,       /* no usable GPU / device id out of range */
    HG_ERR_STATE = 4,           /* call order: image / mesh / prepared frame missing */
    HG_ERR_NOMEM = 5,
    HG_ERR_RANGE = 6            /* a reference-state warp whose map names a triangle without a matrix: the reference's loop throws a
                                   TypeError there (`matrix[0]` of undefined, :1383) */
};

enum { HG_AFFINE = 0, HG_PROJECTIVE = 1 };

typedef struct hg_ctx hg_ctx;

/* Output window of one frame, the reference's (_xOutputOffset, _yOutputOffset, _objectiveWidth, _objectiveHeight).
 * Accepted: up to 2^31 pixels, offsets up to 2^26 in magnitude, at most 65535 frames per set (HG_ERR_INVALID beyond);
 * obj_w / obj_h <= 0 = empty frame. */
typedef struct hg_geom { int32_t x_off, y_off, obj_w, obj_h; } hg_geom;

/* ------------------------------------------------------------------------------------------------ library / context */
int hg_version(void);
int hg_device_count(int *count);
/* Replaces `new Homography()` (:78) for the device side: owns a stream and the device buffers. */
int hg_create(int device_id, hg_ctx **ctx);
/* Same, but launches on a stream owned by the caller (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream). */
int hg_create_on_stream(int device_id, void *hip_stream, hg_ctx **ctx);
void hg_destroy(hg_ctx *ctx);
/* Text of the last error on this ctx (or, with ctx == NULL, of the last failed hg_create / host call of this thread). */
const char *hg_last_error(const hg_ctx *ctx);
/* Waits for the ctx stream; returns the first deferred error of the asynchronous `_device` calls since the last sync.
 * Also settles queued piecewise runs: hg_warp_inverse_piecewise_frames_device calls are queued back to back (up to 63
 * before the library syncs by itself); a frame whose mesh is denser than the fast path's row lists (or whose spans are
 * irregular) is only flagged by the kernel and is redone here, through the materialised map, into the output buffer of
 * the call that flagged it (from the frame set that call was given: a staged copy, newer sets may have gone up since).  Queued
 * calls that wrote overlapping bytes are settled in call order: a later call's frame goes back on top of an earlier call's redo,
 * and an earlier redo is skipped when a later call wrote exactly the same byte range (one buffer reused step after step).  The
 * same holds for queued hg_warp_forward_piecewise_batch_device calls (tile lists over capacity, triangles the tiles cannot bound).
 * A frame is final once hg_sync (or any synchronous call) has returned. */
int hg_sync(hg_ctx *ctx);
/* Device scratch/output helpers so that bindings without a device allocator (Node) can keep frames resident. */
int hg_device_alloc(hg_ctx *ctx, size_t bytes, void **dptr);
int hg_device_free(hg_ctx *ctx, void *dptr);
int hg_copy_to_host(hg_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);
int hg_copy_to_device(hg_ctx *ctx, void *dst_device, const void *src_host, size_t bytes);
/* D2H queued on the ctx stream behind the warps issued so far; dst must stay alive until hg_sync().  A true asynchronous
 * DMA when dst is pinned memory from hg_host_alloc (page-locked, usable by every device). */
int hg_copy_to_host_async(hg_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);
/* The same without settling queued runs first (purely stream-ordered): for callers that queue the copies of several devices
 * before waiting for any (hg_multi_*); frames a fused run only flagged are rewritten by the later hg_sync -- hg_redone_frames
 * tells -- and must then be copied again. */
int hg_enqueue_copy_to_host(hg_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);
/* Host -> device on the context's COPY stream (a second stream owned by the ctx), not ordered with the warp stream until
 * hg_fence_copies(): the upload of frame f + 1's source overlaps frame f's D2H on the warp stream (full-duplex PCIe).  Pageable src:
 * returns once the caller's memory has been read.  hg_fence_copies: everything queued on the warp stream afterwards waits for the
 * uploads queued so far.  (The video loop `for (f) warp(image_f)`, README.md:121-137, through js/Homography.mjs warpBatch({images}).) */
int hg_upload_on_copy_stream(hg_ctx *ctx, void *dst_device, const void *src_host, size_t bytes);
int hg_fence_copies(hg_ctx *ctx);
/* Device -> host on the context's DOWNLOAD stream (a third stream owned by the ctx), ordered behind the warps issued so far and nothing
 * issued later: frame f comes down while frame f + 1 is warped and image f + 2 goes up.  Stream-ordered like hg_enqueue_copy_to_host (a
 * frame a fused run flagged is rewritten by the next hg_sync -- hg_redone_frames tells -- and is then downloaded again by the caller).
 * hg_fence_downloads: the host waits for the downloads queued so far (hg_sync does not wait for them). */
int hg_download_behind_warps(hg_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);
int hg_fence_downloads(hg_ctx *ctx);
/* Everything queued on the ctx stream after this call waits for hip_event (a hipEvent_t recorded on any stream of any device):
 * orders warps behind the caller's own uploads / peer copies without blocking the host. */
int hg_stream_wait_event(hg_ctx *ctx, void *hip_event);
int hg_host_alloc(size_t bytes, void **ptr);
int hg_host_free(void *ptr);
int hg_ctx_device(const hg_ctx *ctx);

/* ------------------------------------------------------------------------------------------------ host-side solves
 * Tiny, run on the host in double precision with the reference's exact operation order; no GPU needed. */
/* affineMatrixFromTriangles :1265-1306 */
int hg_solve_affine(const float src[6], const float dst[6], float out[6]);
/* inverseAffineMatrix :1345-1365 */
int hg_invert_affine(const float m[6], float out[6]);
/* projectiveMatrixFromSquares :1320-1333 + numeric.js solve/LU/LUsolve :1650-1751; out = h0..h7 (h8 == 1) */
int hg_solve_projective(const float src[8], const float dst[8], double out[8]);
/* calculateTransformLimits :1503-1527; kind HG_AFFINE (m[0..5]) / HG_PROJECTIVE (m[0..7]); out = xOff,yOff,objW,objH
 * as doubles exactly as JS computes them (may be NaN/Inf for degenerate matrices). */
int hg_transform_limits(int kind, const double *m, double width, double height, double out[4]);
/* minmaxXYofArray(array, rounded=true) :1558-1589; out = round(minX), round(minY), round(maxX), round(maxY) */
int hg_minmax_xy(const float *points, int n_values, double out[4]);
/* Math.round (ties toward +Infinity), exposed so that bindings share one definition. */
double hg_js_round(double x);
/* Host Delaunay triangulation, where the reference calls `new Delaunator(points).triangles` (:1216-1218 <- :262, :742).
 * points = n_points interleaved x,y.  Writes up to `capacity` triangles (3 vertex ids each) and stores the total count in
 * *n_triangles; pass out_triangles = NULL to query the count (at most 2*n_points).  The algorithm is a restatement of
 * delaunator 5.0.0's (sweep-hull, its seed choice, sort, hull hash and flip order; orientation by an exact-sign predicate), so
 * the LIST is meant to equal delaunator's -- order and diagonals decide pixels where triangles meet -- but delaunator's
 * source is absent from the reference tree and no reference test pins it: "triangulation parity unpinned".  Identical
 * (order included) to js/delaunay.mjs.  HG_ERR_INVALID for non-finite input or a too-small buffer. */
int hg_triangulate(const float *points, int n_points, uint32_t *out_triangles, int capacity, int *n_triangles);

/* ------------------------------------------------------------------------------------------------ source image */
/* setImage, ImageData branch :297-299: uploads RGBA8 (the reference aliases caller memory and re-reads it on every
 * warp(); callers that mutate the buffer call this again). */
int hg_set_image(hg_ctx *ctx, const uint8_t *rgba, int width, int height);
/* Same, source already in GPU memory (aliased, not copied: it must stay alive and unchanged while warps run).
 * This is how a source texture received by an RCCL broadcast is attached without another copy. */
int hg_set_image_device(hg_ctx *ctx, const void *d_rgba, int width, int height);
/* The video case `for (f) { warp(frame_f) }` (README.md:121-137: every warp() gets its own image, setImage :290 per
 * frame): n_images sources of identical size, `stride_bytes` apart in GPU memory (aliased).  Frame f of a frame set
 * (hg_*_set_frames), and frame f of a batch of the forward (scatter-semantics) entry points hg_warp_forward_*_batch_device, then
 * reads image f % n_images; n_images == 1 is hg_set_image_device. */
int hg_set_images_device(hg_ctx *ctx, const void *d_rgba, int width, int height, int n_images, size_t stride_bytes);

/* ------------------------------------------------------------------------------------------------ affine / projective
 * _inverseGeometricWarp pixel loop :997-1011 (+ applyAffineTransformToPoint :1382 / applyProjectiveTransformToPoint
 * :1401).  `m` is the INVERSE matrix the reference obtains at :994 by re-solving with the point sets swapped
 * (hg_solve_affine(dst, src) widened to double, or hg_solve_projective(dst, src)): 6 resp. 8 doubles.
 * Output: 4*obj_w*obj_h bytes, every pixel written (0 where the reference leaves the zero-initialised buffer). */
int hg_warp_inverse_geometric(hg_ctx *ctx, int kind, const double *m, hg_geom geom, uint8_t *out_host);
int hg_warp_inverse_geometric_device(hg_ctx *ctx, int kind, const double *m, hg_geom geom, void *d_out);
/* The caller loop `setDestinyPoints(dst_f); warp()` (test/benchmark.js:107-110) for F frames in one launch:
 * m = F x 8 doubles (affine uses the first 6 of each 8), geoms = F windows, out_offsets = F byte offsets into d_out
 * (NULL: packed as hg_pack_offsets does).  set_frames uploads the per-frame inputs (kept until replaced);
 * frames_device runs all uploaded frames; batch_device = both. */
int hg_geometric_set_frames(hg_ctx *ctx, int kind, const double *m, const hg_geom *geoms, const size_t *out_offsets, int n_frames);
/* Same frame set given as POINT SETS: the reference obtains the inverse matrix at the head of every inverse warp by
 * re-solving with the swapped sets (:994 calculateTransformMatrix(dstPoints, srcPoints) -> projectiveMatrixFromSquares
 * :1320-1333 + numeric.js LU/LUsolve :1650-1751, or affineMatrixFromTriangles :1265-1306).  from / to = n_frames x 4 (or
 * 3) points x,y float32; the matrix of frame f maps from[f] -> to[f] (pass dst, src for the inverse warp).  Every
 * hg_warp_inverse_geometric_frames_device then runs the solves ON THE DEVICE (one lane per frame, the LU in its exact
 * operation order; bit patterns equal to hg_solve_projective / hg_solve_affine) followed by the pixel kernel. */
int hg_geometric_set_frames_points(hg_ctx *ctx, int kind, const float *from, const float *to, const hg_geom *geoms,
                                   const size_t *out_offsets, int n_frames);
/* Parity tap: the n_frames x 8 doubles the pixel kernel uses (device-solved when the frames were given as point sets). */
int hg_get_geometric_matrices(hg_ctx *ctx, double *out, int n_frames);
int hg_warp_inverse_geometric_frames_device(hg_ctx *ctx, void *d_out);
int hg_warp_inverse_geometric_batch_device(hg_ctx *ctx, int kind, const double *m, const hg_geom *geoms,
                                           const size_t *out_offsets, int n_frames, void *d_out);
/* Packed output layout for n frames: offsets[i] = start of frame i (256-byte aligned), *total = bytes needed. */
int hg_pack_offsets(const hg_geom *geoms, int n_frames, size_t *offsets, size_t *total);

/* ------------------------------------------------------------------------------------------------ piecewise affine
 * Source side of the mesh, kept across frames like the reference's cached _srcPoints/_triangles/_minSrcX/_minSrcY
 * (:742, :758).  min_src_x/min_src_y are the rounded source-point bbox minimum used by the bounds test :1047.
 * Coordinates (source and destiny) may be NaN but not infinite or beyond 2^24 in magnitude: HG_ERR_INVALID (the
 * reference's fillTriangle row loop would run for as many rows as the triangle is tall -- forever for Infinity). */
int hg_piecewise_set_mesh(hg_ctx *ctx, const float *src_points, int n_points, const uint32_t *triangles, int n_triangles,
                          int min_src_x, int min_src_y);
/* Per-frame destination side = what setDestinyPoints + the head of _inversePiecewiseAffineWarp recompute every frame:
 * _calculatePiecewiseAffineTransformMatrices :785-804 (affineMatrixFromTriangles per triangle, f32),
 * inverseAffineMatrix per triangle :1036-1038, and the edge equations / row ranges of fillTriangle :1111-1151.
 * All on the device; dst_points are n_points x,y float32 in pixel coordinates. */
int hg_piecewise_prepare(hg_ctx *ctx, const float *dst_points, hg_geom geom);
/* _inversePiecewiseAffineWarp :1029-1058 for the prepared frame.  The triangle-index map of
 * _buildInverseTrianglesCorrespondencesMatrix :845-861 is not materialised: each output row's triangle spans
 * (fillTriangle/predictXLimits, with TypedArray.fill index semantics) are rebuilt in LDS and resolved per pixel with
 * "largest covering id wins", which equals the reference's sequential overwrite. */
int hg_warp_inverse_piecewise(hg_ctx *ctx, uint8_t *out_host);
int hg_warp_inverse_piecewise_device(hg_ctx *ctx, void *d_out);
/* F frames (F destination point sets on the mesh set above) in one pass: dst_points = F x n_points x 2 float32.
 * set_frames uploads points + windows (kept until replaced; hg_piecewise_prepare == set_frames with one frame);
 * frames_device runs the per-frame solves and the warp for all uploaded frames; batch_device = both.
 * set_frames does NOT wait for the GPU (the reference's loop calls setDestinyPoints before every warp, test/benchmark.js:107-110,
 * Homography.js:337-380): the caller's arrays are copied into page-locked staging owned by the ctx before it returns, the
 * uploads are ordered behind the queued runs on the ctx stream, and every queued run keeps the staged copy of the set it
 * warped, so that frames it flagged can still be redone by hg_sync after newer sets went up.  Up to 63 set / run pairs
 * queue between two hg_sync calls; the 64th settles the queue by itself. */
int hg_piecewise_set_frames(hg_ctx *ctx, const float *dst_points, const hg_geom *geoms, const size_t *out_offsets, int n_frames);
int hg_warp_inverse_piecewise_frames_device(hg_ctx *ctx, void *d_out);
int hg_warp_inverse_piecewise_batch_device(hg_ctx *ctx, const float *dst_points, const hg_geom *geoms,
                                           const size_t *out_offsets, int n_frames, void *d_out);
/* Parity taps (debug / tests): the Int16Array map the reference would have built for the prepared frame
 * (len = obj_w*obj_h), via the materialising kernel (atomicMax rasteriser) or via the fused kernel's own lookup;
 * and the per-triangle forward / inverse matrices (n_triangles x 6 float32 each; either may be NULL). */
int hg_get_tri_map(hg_ctx *ctx, int16_t *out, size_t len);
int hg_get_tri_map_fused(hg_ctx *ctx, int16_t *out, size_t len);
int hg_get_matrices(hg_ctx *ctx, float *fwd, float *inv);
/* Same warp through the materialised map (map build kernel + map-reading warp kernel).  Always available; it is also
 * what the library itself re-runs for a frame whose rows overflow the fused kernel's LDS span list. */
int hg_warp_inverse_piecewise_via_map(hg_ctx *ctx, uint8_t *out_host);

/* ------------------------------------------------------------------------------------------------ forward (scatter) paths
 * What warp() dispatches to when the output is not larger than the input (:421, :426).  `m` is the FORWARD matrix
 * (_transformMatrix).  Sequential "last writer in raster order wins" is reproduced deterministically (atomicMax of the
 * raster rank per output pixel, then a gather).  Synchronous, host output of 4*obj_w*obj_h bytes. */
/* _geometricWarp :911-932.  Limits of the forward paths (HG_ERR_INVALID beyond): source image resp. source-point bounding box
 * below 2^31 pixels and at most 65535 rows; windows as for the inverse paths. */
int hg_warp_forward_geometric(hg_ctx *ctx, int kind, const double *m, hg_geom geom, uint8_t *out_host);
/* Same, asynchronous into GPU memory; batch = n frames (m = n x 8 doubles) back to back, frame f at out_offsets[f]. */
int hg_warp_forward_geometric_device(hg_ctx *ctx, int kind, const double *m, hg_geom geom, void *d_out);
int hg_warp_forward_geometric_batch_device(hg_ctx *ctx, int kind, const double *m, const hg_geom *geoms, const size_t *out_offsets,
                                           int n_frames, void *d_out);
/* _piecewiseAffineWarp :948-972 on the mesh of hg_piecewise_set_mesh (whose min_src_x/y are the loop origin);
 * max_src_x/y = rounded source-point bbox maximum (:758); the forward triangle map :817-832 is rebuilt on the device. */
int hg_warp_forward_piecewise(hg_ctx *ctx, const float *dst_points, int max_src_x, int max_src_y, hg_geom geom, uint8_t *out_host);
/* Same, asynchronous into GPU memory; batch = the caller loop `setDestinyPoints(d_f); warp()` for n destination point sets
 * (dst_points = n_frames x n_points x,y) when warp() takes the forward path (:421).  Like the inverse `_device` entry points
 * the frames are final after hg_sync (or any hg_copy_to_host): a frame the tile-binned kernels flagged is redone there. */
int hg_warp_forward_piecewise_device(hg_ctx *ctx, const float *dst_points, int max_src_x, int max_src_y, hg_geom geom, void *d_out);
int hg_warp_forward_piecewise_batch_device(hg_ctx *ctx, const float *dst_points, int max_src_x, int max_src_y, const hg_geom *geoms,
                                           const size_t *out_offsets, int n_frames, void *d_out);

/* ------------------------------------------------------------------------------------------------ reference-state forms
 * The reference caches two things across calls and its loops read them AS THEY STAND:
 *   - `_piecewiseMatrices` (:769): solved at setDestinyPoints, NOT invalidated by setTriangles (:519) nor -- once a map exists -- by
 *     setSourcePoints (:252-255);
 *   - `_trianglesCorrespondencesMatrix`: ONE field for the forward map (:819-820, source points over the source bounding box) and the
 *     inverse map (:847-848, destiny points over the output window).  After an inverse warp, _piecewiseAffineWarp :957 indexes the
 *     stale inverse map with forward indices ((y - minSrcY) * (maxSrcX - minSrcX) + (x - minSrcX)); cells past its end read
 *     `undefined` and are skipped.
 * A binding that mirrors those caches (js/Homography.mjs does, as value snapshots) uses the entry points above whenever both are
 * current -- the only state a fresh instance can be in -- and these two otherwise.  They take the cached state explicitly: the
 * FORWARD matrices as they were last solved (6 floats per triangle, hg_solve_affine_triangles of the snapshot) and the definition
 * of the map the field holds (the point set + triangles fillTriangle rasterised, matrix_width, height = length / width, yOffset).
 * Exact for any input (materialised map: atomicMax rasteriser + the pixel loop reading it), synchronous, host output of
 * 4 * obj_w * obj_h bytes.  HG_ERR_RANGE when a cell the loop reads holds an id >= n_mats (JS: TypeError at that pixel); the forward
 * form makes that check for a blank window too (the reference's loop runs over the source bounding box whatever the output size).
 * LIMITS of these two forms (HG_ERR_INVALID, a string error in the class, where the reference would still produce pixels or fail
 * differently -- none of them reachable from finite pixel coordinates of images that fit the other entry points): a map coordinate
 * that is infinite or beyond 2^24 in magnitude (NaN is legal: such a triangle rasterises nothing); |map y_off| > 2^26; a map of 2^31
 * cells or more; forward form: a source-point bounding box of 2^31 pixels or more, or taller than 65 535 rows.  Triangle ids are the
 * reference's Int16Array values: with more than 32 767 triangles they wrap (id 32 768 reads back as -32 768 = "no triangle"),
 * reproduced as such. */
typedef struct hg_tri_map_def {
    const float *points; int n_points;              /* interleaved x,y of the point set the map was rasterised from */
    const uint32_t *triangles; int n_triangles;
    int32_t width, height, y_off;                   /* fillTriangle's matrix_width, Int16Array length / width, yOffset (:1111) */
} hg_tri_map_def;
/* affineMatrixFromTriangles for every triangle of a mesh (:785-804; host, no GPU): out = 6 floats per triangle; a vertex id beyond
 * n_points reads `undefined` -> NaN like the reference's Float32Array scratch. */
int hg_solve_affine_triangles(const float *src_points, const float *dst_points, int n_points, const uint32_t *triangles, int n_triangles, float *out);
/* _inversePiecewiseAffineWarp :1029-1058 with stale matrices: the map is always the one of the current destiny points over the
 * current window (:1033), so map->width / height / y_off must equal geom's obj_w / obj_h / y_off. */
int hg_warp_inverse_piecewise_state(hg_ctx *ctx, const float *fwd_mats, int n_mats, const hg_tri_map_def *map, int min_src_x, int min_src_y,
                                    hg_geom geom, uint8_t *out_host);
/* _piecewiseAffineWarp :948-972 over whatever map the shared field holds. */
int hg_warp_forward_piecewise_state(hg_ctx *ctx, const float *fwd_mats, int n_mats, const hg_tri_map_def *map, int min_src_x, int min_src_y,
                                    int max_src_x, int max_src_y, hg_geom geom, uint8_t *out_host);

/* ------------------------------------------------------------------------------------------------ several GPUs, one host thread
 * The caller loop `for (f) { setDestinyPoints(dst_f); warp(); }` (test/benchmark.js:107-110) spread over the devices of one
 * node for hosts that cannot run one process per GPU (the Node binding).  Frames are independent: device i
 * of G warps the contiguous block hg_multi_partition() assigns to it, no data-path collective; the shared source texture is
 * fanned out once per hg_multi_set_image over xGMI peer copies (scatter of 1/G slices from device 0, then an all-gather
 * between the peers, so that every point-to-point link carries 1/G of the image instead of one link carrying all of it).
 * A device id may be listed more than once (several contexts on one GPU). */
typedef struct hg_multi hg_multi;
int hg_multi_create(const int *device_ids, int n_devices, hg_multi **multi);
void hg_multi_destroy(hg_multi *multi);
const char *hg_multi_last_error(const hg_multi *multi);
int hg_multi_device_count(const hg_multi *multi);
hg_ctx *hg_multi_ctx(hg_multi *multi, int index);                       /* the per-device context (options, taps) */
/* Peer access is checked and enabled pair by pair at hg_multi_create.  hg_multi_peer_note: "" when every pair of distinct devices
 * has it, else a text listing the pairs without ("no peer access: 2->3, 3->2"); hg_multi_peer_access: 1 / 0 for one ordered pair of
 * indices into the device list (-1: bad index).  A pair without access only re-routes ITS copies of the source fan-out. */
const char *hg_multi_peer_note(const hg_multi *multi);
int hg_multi_peer_access(const hg_multi *multi, int from_index, int to_index);
/* Pure function (no GPU): the copy plan of the source fan-out for n_devices devices and the access matrix access[p * n_devices + q]
 * (non-zero: device q copies straight out of device p's memory).  Returns the number of copies and writes up to max_ops of them, 4
 * ints each: {destination index, source index or -1 = the caller's host buffer, slice, phase 0 scatter / 1 all-gather}.  Scatter:
 * slice q goes to device q from device 0; all-gather: q takes slice p from its owner p, else from device 0 (which holds every
 * slice), else from the host.  With full access every ordered pair of devices carries exactly one 1/n slice. */
int hg_multi_plan_fanout(int n_devices, const uint8_t *access, int32_t *ops, int max_ops);
/* Pure function: frames [*first, *first + *count) of n_frames belong to device `index` of n_devices (sizes differ by <= 1). */
int hg_multi_partition(int n_frames, int n_devices, int index, int *first, int *count);
/* Returns once the host buffer has been read (H2D into device 0, plus the slices of devices no other device can serve).  The fan-out
 * between the devices is NOT waited for: every device's warp stream waits for the event "whole image here", so device 0's
 * first frames overlap the scatter + all-gather. */
int hg_multi_set_image(hg_multi *multi, const uint8_t *rgba, int width, int height);
int hg_multi_piecewise_set_mesh(hg_multi *multi, const float *src_points, int n_points, const uint32_t *triangles, int n_triangles,
                                int min_src_x, int min_src_y);
/* dst_points = n_frames x n_points x,y; out_host = NULL (frames stay on their devices: hg_multi_frame) or n_frames host
 * pointers of 4*obj_w*obj_h bytes each (pinned memory from hg_host_alloc lets the devices' copies overlap).  Synchronous. */
int hg_multi_warp_piecewise_batch(hg_multi *multi, const float *dst_points, const hg_geom *geoms, int n_frames, uint8_t *const *out_host);
/* The same for affine / projective frames given as point sets (see hg_geometric_set_frames_points): the matrix of frame f maps
 * from[f] -> to[f] (3 or 4 points each) and is solved on the device that warps the frame. */
int hg_multi_warp_geometric_batch(hg_multi *multi, int kind, const float *from, const float *to, const hg_geom *geoms, int n_frames,
                                  uint8_t *const *out_host);
/* One source per frame -- the video loop `for (f) { warp(frame_f) }` (README.md:121-137), SURVEY.md §8e's "replicas only" branch:
 * images[f] = width x height RGBA8 in host memory for frame f; every device uploads the images of ITS block only, there is no
 * exchange between devices at all.  Afterwards the contexts read those per-frame sources: call hg_multi_set_image again
 * before the next shared-source batch. */
int hg_multi_warp_piecewise_batch_images(hg_multi *multi, const float *dst_points, const hg_geom *geoms, int n_frames,
                                         const uint8_t *const *images, int width, int height, uint8_t *const *out_host);
int hg_multi_warp_geometric_batch_images(hg_multi *multi, int kind, const float *from, const float *to, const hg_geom *geoms, int n_frames,
                                         const uint8_t *const *images, int width, int height, uint8_t *const *out_host);
/* Where frame f of the last batch lives: index into the device list, device pointer, byte size (any of them may be NULL). */
int hg_multi_frame(hg_multi *multi, int frame, int *device_index, void **d_ptr, size_t *bytes);

/* Which kernel produced the last fused piecewise warp of this ctx (tests / profiling): 0 = none yet, 1 = k_pw_rows with
 * 4-row groups, 2 = k_pw_rows one row per workgroup, 3 = k_pw_patch (dense sheared meshes), 4 = k_pw_fused (general),
 * 5 = k_pw_tile (8-row tiles whose gathers follow the source rows: one source per frame, steeply sheared or very dense meshes). */
int hg_last_piecewise_kernel(hg_ctx *ctx);
/* ... and which template instantiation of it (tools/census.py: the census of what the layout policy really picks): kind * 100000 +
 * (512-slot rows) * 10000 + windows / blocks per phase * 1000 + (8-byte row entries) * 100 + (bounds on the high dwords) * 10 + self-span
 * form; kind 1 k_pw_rows, 3 k_pw_rows_s80, 4 k_pw_patch, 5 k_pw_tile, 6 k_pw_fused, 8 k_pw_patch with records in global memory. */
int hg_last_piecewise_variant(hg_ctx *ctx);
/* 1 if that run's row workgroups evaluated their own spans (k_tri_setup + k_pw_rows<SELF>: no row lists, no slot atomics, option
 * "self_spans"), 0 if they read the per-output-row span lists of k_tri_spans. */
int hg_last_piecewise_self(hg_ctx *ctx);
/* Which kernels ran the last forward warp: 0 = none yet, 1 = scatter (atomicMax on a winner buffer) + gather, 2 = the
 * tile-binned gather (output tiles gather their source pixels, winners resolved in LDS, one or two launches for the whole
 * batch): k_fwd_tiles for affine / projective matrices that pass the admissibility bounds, k_fwd_pw_bins + k_fwd_pw_tiles
 * for piecewise meshes (frames the device cannot bound are flagged and redone through 1 by hg_sync; hg_redone_frames counts
 * them).  Option "fwd_tiles": -1 by size and mesh density (default), 0 never, 1 whenever admissible. */
int hg_last_forward_kernel(hg_ctx *ctx);
/* Host-side admission test of k_fwd_tiles (no GPU needed): 0 = this forward matrix (6 or 8 doubles) / source size / window
 * goes through scatter + gather; 1 = admissible; 2 = admissible and the inverse is trusted for the per-tile source row range.
 * Bounds: DESIGN.md §4.7 / EXPERIMENTS.md (matrix magnitudes, projective denominator >= 1e-2 at the source corners, no source pixel more than
 * 30 columns outside the window, window at least 64 wide). */
int hg_forward_tiles_admissible(int kind, const double *m, int W, int H, hg_geom geom);
/* Status word of the last frame a fused piecewise run flagged (0 = none since hg_create; diagnostics): bit 0 = a triangle with
 * non-finite / absurd vertices, bit 1 = a kernel's limits (more spans in a row than its lists or LDS blocks hold, more candidate triangles
 * than a workgroup's list); with bit 1, bits 4.. say which limit of k_pw_rows<SELF> / k_pw_patch<SELF> / k_pw_tile and the count. */
int hg_last_piecewise_flag(hg_ctx *ctx);
/* Frames the fused kernels only flagged (row lists / kernel limits exceeded, irregular spans) and hg_sync redid through the
 * materialised map, since the ctx was created (tests / profiling: a steady-state workload should show 0). */
long hg_redone_frames(hg_ctx *ctx);
/* Host-side walks over every triangle of a frame set (the layout estimate of hg_piecewise_set_frames) since the ctx was created:
 * a caller that uploads fresh points of one mesh and window shape per step should see this stand still. */
long hg_layout_walks(hg_ctx *ctx);
/* Layout knobs of the piecewise fast path; they never change results (tests run the parity suite under each setting), only
 * which kernel layout the next hg_piecewise_set_frames picks:
 *   "min_row_groups" (default 1152): frame sets with fewer 4-row groups run one row per workgroup on row lists (k_pw_patch sets keep
 *           the self-span form down to half of it);
 *   "patch" (default -1 = by estimate: rows of more than 56 spans in a frame set that fits the kernel, and every set that fits it when
 *           each frame reads its own source): 0 never use k_pw_patch, 1 use it whenever the frame width allows, 2 the same in its
 *           global-record variant;
 *   "phase" (default -1 = 2 for a shared source -- 4 when the rows carry 3 or more spans per window --, 1 with one source per frame):
 *           windows per gather/store phase of k_pw_rows on ROW LISTS, 1, 2 or 4 (the self-span form always takes 4, k_pw_patch 8 column
 *           blocks: the other depths were never picked by the policy and are gone since round 6);
 *   "geo_windows" (default 8): 256-pixel windows per wave of the affine / projective kernel, 1, 2, 4 or 8;
 *   "self_spans" (default -1 = by policy; 1 = whenever eligible: sparse meshes of up to 8192 triangles; 0 never): no span
 *           producer kernel and no per-output-row span lists -- k_tri_setup writes every triangle's edge equations, inverse matrix
 *           and row reach, and each row workgroup of the warp kernel picks the triangles that reach its rows and evaluates
 *           predictXLimits + the fill() indices for exactly those rows in its prologue.  Bit-identical;
 *   "hi_bounds" (default 1): the source-bounds tests of the pixel loops (:1047, :1001) as 32-bit compares on the high dwords
 *           of the rounded coordinates (exact whenever the source window starts at >= 0 and ends below 2^20; the kernels
 *           fall back to the fp64 compares by themselves otherwise), 0 = always the fp64 compares;
 *   "compact" (default -1 = by estimate: dense rows, or whenever k_pw_patch reads the lists): 1 / 0: 8-byte span-list entries (the
 *           consumer fetches the matrix by triangle id) / 32-byte entries carrying the matrix;
 *   "tri_group" (default -1 = by mesh size: meshes of >= 384 triangles in sets of >= 2048 (frame, triangle) pairs): 16 or 64 (any
 *           other non-zero value = 16): the span producer takes that many triangles per workgroup and solves them one per lane
 *           (k_tri_spans_grouped) instead of one triangle per workgroup whose waves all repeat its solves (k_tri_spans, the
 *           lower-latency choice for a single frame and for sparse meshes); 0: never;
 *   "xcc_rotate" (default -1 = by estimate): 1: XCD x walks row band (x + frame) mod XCCs instead of band x -- even load where
 *           rows differ in cost or the frames share no source; 0: fixed bands (a shared source's band stays in that XCD's L2);
 *   "sub_bands" (default -1 = by the source's size: as many as bring a sub-band's share of the source to ~2.2 MB -- 2 for a 4K source, none for
 *           1080p; 0 / 1 none; 2..64): with ONE source shared by several frames and fixed bands, the rows of a frame are cut into sub-bands that are
 *           dealt to the XCDs round robin, and an XCD takes all frames of one of its sub-bands before the next: the slice of the source a sub-band
 *           reads then stays in that XCD's 4 MiB L2 from frame to frame (k_pw_rows / k_pw_patch / k_pw_tile);
 *   "xcc" (default: hipDeviceAttributeNumberOfXccs of the device, 8 on an unpartitioned MI355X): number of XCCs the
 *           block id -> row band mapping of the warp kernels assumes; a power of two in 1..64;
 *   "upload_kernel" (default -1 = on): frame-set blocks of up to 1 MB go from their page-locked staging slot to the device by a small kernel
 *           that reads host memory instead of a stream-ordered hipMemcpyAsync (whose copy-engine start-up cost 10-15 us per set); 0: always
 *           the copy engine;
 *   "safe_spans" (default -1 = rows of fewer than 3 spans per 256-pixel window; 1 / 0 force / forbid): k_pw_rows flags every span whose two end
 *           pixels pass the source bounds test :1047 (then every pixel between them does: the affine coordinates are monotone along a row) and
 *           runs windows made only of such spans without the per-pixel test, Math.round with one add per coordinate;
 *           k_pw_patch<SELF> does the same per 64-pixel x 4-row block (on unless the option says 0);
 *   "tile" (default -1 = wherever k_pw_patch would evaluate its own spans and either every frame reads its own source or, on a shared
 *           source, the mesh is steeply sheared (estimate >= 0.3) or packs more than two spans into a 64-pixel block -- and the host's
 *           estimate says a tile row holds its spans; 1 = whenever k_pw_patch<SELF> would run; 0 never): k_pw_tile instead of k_pw_patch -- 8 x 2048-pixel tiles, gathers in 8-pixel runs along the source
 *           rows; a tile beyond its limits (96 spans per row and tile, 128 triangle pieces) flags its frame, hg_sync
 *           redoes it and the mesh goes back to k_pw_patch;
 *   "fwd_tiles": see hg_last_forward_kernel.
 * Unknown keys are refused (HG_ERR_INVALID). */
int hg_set_option(hg_ctx *ctx, const char *key, int value);
/* Number of XCCs the ctx maps row bands to (what hg_create read from the device, or the "xcc" option). */
int hg_xcc_count(const hg_ctx *ctx);
/* Host-side proof obligation of that division (no GPU needed): 1 if every pixel of the window `geom` under the inverse
 * projective matrix m[8] keeps numerators and denominator in the plain range (entries 0 or in [2^-100, 2^100], coordinates
 * below 2^28, denominator of one sign and in [2^-100, 2^130] at the four corners), else 0 (-> IEEE divides in the kernel). */
int hg_projective_plain_range(const double *m, hg_geom geom);
/* Host-side proof obligation of the piecewise row kernel's one-fma form (no GPU needed): 1 if, for every integer pixel (x, y) of the window
 * `geom`, (m0*x) + (m2*y) + m4 and (m1*x) + (m3*y) + m5 of applyAffineTransformToPoint :1382-1385 -- inv[6] = an inverse matrix, f32 values --
 * round at most once, so that fma(m0, x, (m2*y) + m4) has the same bits (both partial sums exactly representable; hg_math.h
 * affine_fusable); else 0 (-> the kernel keeps the two roundings).  k_tri_setup evaluates the same predicate per (frame, triangle). */
int hg_affine_one_fma_form(const float inv[6], hg_geom geom);
/* Self-test of the projective kernels' shared-reciprocal division against IEEE division on `samples` pseudo-random
 * operand triples drawn from the range the host admits it for; *mismatches must come back 0. */
int hg_selftest_division(hg_ctx *ctx, uint64_t samples, uint64_t seed, uint64_t *mismatches);
/* ------------------------------------------------------------------------------------------------ measurement aid
 * hipEvent pairs recorded on the ctx stream around each launch of the dominant kernel (the fused piecewise kernel or
 * the geometric kernel; not the tiny per-triangle setup).  hg_set_timing(ctx, 1) enables it and resets the counters;
 * hg_last_kernel_ms = duration of the most recent launch; hg_kernel_ms_stats = sum over (up to the last 256) launches
 * since the reset and how many that is.  Both wait for the stream. */
int hg_set_timing(hg_ctx *ctx, int enabled);
int hg_last_kernel_ms(hg_ctx *ctx, float *ms);
int hg_kernel_ms_stats(hg_ctx *ctx, double *total_ms, int *launches);

#ifdef __cplusplus
}
#endif
#endif /* HGWARP_H */
