/*
 * hg_oracle.h -- CPU restatement (plain C) of the reference's per-pixel warp path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it, and only as the checker / baseline.  The shipped path is the HIP
 * library (homography.js_amd/csrc) and it has no CPU fallback.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py) against golden vectors
 * produced by running the reference's own Homography.js (v1.8.0) in the build container
 * (tests/golden/gen_golden.mjs) and against the reference's one known-answer fixture
 * (test/testImgLogoBlack.png -> test/transformedImage.png).  One thing is NOT pinned: the Delaunay triangulation
 * (delaunator@5.0.0, source absent from the reference tree) -- triangles are always an explicit input here.
 *
 * All arithmetic that the reference does on JS Numbers is done on C doubles with contraction disabled
 * (-ffp-contract=off), in the reference's written operation order.  file:line citations are into
 * /root/reference/Homography.js.
 */
#ifndef HG_ORACLE_H
#define HG_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Math.round (ECMA-262): nearest integer, ties toward +Infinity.  Used at :924 :962 :1005 :1048 :1124 :1525 :1585 */
double hgo_js_round(double x);

/* affineMatrixFromTriangles :1265-1306  (double math, result rounded to f32; layout x'=a*x+c*y+e, y'=b*x+d*y+f) */
void hgo_affine_from_triangles(const float src[6], const float dst[6], float out[6]);
/* inverseAffineMatrix :1345-1365 */
void hgo_inverse_affine(const float m[6], float out[6]);
/* projectiveMatrixFromSquares :1320-1333 + numeric.js solve/LU/LUsolve :1650-1751 (out = h0..h7, h8 == 1) */
void hgo_projective_from_squares(const float src[8], const float dst[8], double out[8]);

/* fillTriangle :1111-1126 (+ defineTriangleLineEquations :1141-1151, predictXLimits :1172-1197) into an Int16Array
 * of `len` cells, with TypedArray.prototype.fill index semantics. */
void hgo_fill_triangle(const float tri[6], double idx, double matrix_width, double y_offset, int16_t *map, int64_t len);
/* _buildInverseTrianglesCorrespondencesMatrix :845-861 / _buildTrianglesCorrespondencesMatrix :817-832:
 * map.fill(-1) then fillTriangle for every triangle in ascending order. */
void hgo_build_tri_map(const float *points, const uint32_t *tris, int n_tris, double matrix_width, double y_offset,
                       int16_t *map, int64_t len);
/* _calculatePiecewiseAffineTransformMatrices :785-804 -> n_tris x 6 f32 */
void hgo_piecewise_matrices(const float *src_pts, const float *dst_pts, const uint32_t *tris, int n_tris, float *fwd);

/* calculateTransformLimits :1503-1527.  kind 0 = affine (m[0..5], f32-valued), 1 = projective (m[0..7]).
 * out = {xOff, yOff, objW, objH} as doubles (may be NaN/Inf exactly as in JS). */
void hgo_transform_limits(int kind, const double *m, double width, double height, double out[4]);
/* minmaxXYofArray(array, rounded=true) :1558-1589 -> {minX, minY, maxX, maxY} rounded */
void hgo_minmax_xy(const float *pts, int n_values, double out[4]);

/* _inverseGeometricWarp pixel loop :997-1011.  kind 0 affine / 1 projective, m = the INVERSE matrix
 * (calculateTransformMatrix(transform, dst, src), :994).  out must hold 4*objW*objH bytes; it is zeroed first (:991). */
void hgo_warp_inverse_geometric(int kind, const double *m, const uint8_t *image, int W, int H,
                                int xOff, int yOff, int objW, int objH, uint8_t *out);
/* _inversePiecewiseAffineWarp pixel loop :1042-1056 given the int16 map and the inverse f32 matrices. */
void hgo_warp_inverse_piecewise_loop(const int16_t *map, const float *inv, int n_tris, const uint8_t *image, int W, int H,
                                     int minSrcX, int minSrcY, int xOff, int yOff, int objW, int objH, uint8_t *out);
/* Whole _inversePiecewiseAffineWarp :1029-1058 as the reference runs it per frame: forward matrices (:785-804),
 * inverse map (:845-861), inverse matrices (:1036-1038), pixel loop.  map_out (objW*objH int16), fwd_out/inv_out
 * (n_tris*6 f32) are optional taps (may be NULL). */
void hgo_warp_inverse_piecewise(const float *src_pts, const float *dst_pts, const uint32_t *tris, int n_tris,
                                const uint8_t *image, int W, int H, int minSrcX, int minSrcY,
                                int xOff, int yOff, int objW, int objH,
                                uint8_t *out, int16_t *map_out, float *fwd_out, float *inv_out);

/* Forward scatter paths: _geometricWarp :911-932 and _piecewiseAffineWarp :948-972. */
void hgo_warp_forward_geometric(int kind, const double *m, const uint8_t *image, int W, int H,
                                int xOff, int yOff, int objW, int objH, uint8_t *out);
void hgo_warp_forward_piecewise(const int16_t *fwd_map, const float *fwd, int n_tris, const uint8_t *image, int W, int H,
                                int minSrcX, int minSrcY, int maxSrcX, int maxSrcY,
                                int xOff, int yOff, int objW, int objH, uint8_t *out);

/* Synthetic RGBA source of SURVEY.md §8d: s = s*1664525 + 1013904223 (mod 2^32), byte = s >> 24. */
void hgo_lcg_image(uint8_t *data, size_t n_bytes, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif
