// hg_math.h -- exact-arithmetic building blocks shared by host (C ABI solves) and device (gfx950 kernels).
//
// Everything here restates arithmetic the reference performs on JS Numbers (IEEE doubles, no FMA contraction);
// the library is compiled with -ffp-contract=off so each * and + rounds separately in the written order.
// Citations are file:line into the reference's Homography.js (v1.8.0).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define HG_HD __host__ __device__ __forceinline__

namespace hg {

// ---------------------------------------------------------------------------------------------- JS number semantics

// Math.round: nearest integer, ties toward +Infinity (not floor(x+0.5): that fails for 0.49999999999999994).
HG_HD double js_round(double x)
{
    if (!(fabs(x) < 4503599627370496.0)) return x;      // NaN, +-Inf, |x| >= 2^52 are already integral
    double r = floor(x);
    if (x - r >= 0.5) r += 1.0;                           // x - floor(x) is exact
    return r;
}

// ToInt32 (`~~v`): truncate, wrap modulo 2^32; NaN/Inf -> 0.
HG_HD int32_t js_to_int32(double x)
{
    if (!(fabs(x) < INFINITY)) return 0;
    double t = trunc(x);
    if (fabs(t) < 2147483648.0) return (int32_t)t;
    double m = fmod(t, 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (int32_t)(uint32_t)m;
}

HG_HD double js_min2(double a, double b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }
HG_HD double js_max2(double a, double b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }

// TypedArray.prototype.fill relative index (ToIntegerOrInfinity; negative counts from the end; clamp to [0,len]).
HG_HD int64_t js_fill_index(double v, int64_t len)
{
    if (v != v) v = 0.0;
    if (v == -INFINITY) return 0;
    if (v == INFINITY) return len;
    v = trunc(v);
    if (v < 0) { double k = (double)len + v; return k < 0 ? 0 : (int64_t)k; }
    return v > (double)len ? len : (int64_t)v;
}

// ---------------------------------------------------------------------------------------------- per-triangle solves

// affineMatrixFromTriangles :1265-1306.  s,d = [x0,y0,x1,y1,x2,y2] (f32 values); out = [a,b,c,d,e,f] rounded to f32.
HG_HD void solve_affine(const float *s, const float *d, float *out)
{
    const double srcE = s[4], srcF = s[5];
    const double srcA = s[0] - srcE, srcB = s[1] - srcF, srcC = s[2] - srcE, srcD = s[3] - srcF;
    const double dstE = d[4], dstF = d[5];
    const double dstA = d[0] - dstE, dstB = d[1] - dstF, dstC = d[2] - dstE, dstD = d[3] - dstF;
    const double den = srcA * srcD - srcB * srcC;        // :1287
    const double nden = -den;
    const double iA = srcD / den;                         // :1289-1294 (x / -den, never a reciprocal multiply)
    const double iB = srcB / nden;
    const double iC = srcC / nden;
    const double iD = srcA / den;
    const double iE = (srcD * srcE - srcC * srcF) / nden;
    const double iF = (srcB * srcE - srcA * srcF) / den;
    out[0] = (float)((dstA * iA) + (dstC * iB));          // :1297-1304, stored through a Float32Array
    out[1] = (float)((dstB * iA) + (dstD * iB));
    out[2] = (float)((dstA * iC) + (dstC * iD));
    out[3] = (float)((dstB * iC) + (dstD * iD));
    out[4] = (float)((dstA * iE) + (dstC * iF) + dstE);
    out[5] = (float)((dstB * iE) + (dstD * iF) + dstF);
}

// inverseAffineMatrix :1345-1365
HG_HD void invert_affine(const float *m, float *out)
{
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5];
    const double den = a * d - b * c;
    const double nden = -den;
    out[0] = (float)(d / den);
    out[1] = (float)(b / nden);
    out[2] = (float)(c / nden);
    out[3] = (float)(a / den);
    out[4] = (float)((d * e - c * f) / nden);
    out[5] = (float)((b * e - a * f) / den);
}

// projectiveMatrixFromSquares :1320-1333 + numeric.js solve = LUsolve(LU(A)) :1650-1751, in registers: every loop has
// constant bounds and is fully unrolled, so all array indices are compile-time constants (no scratch memory on the GPU);
// the data-dependent row exchange of the partial pivoting becomes conditional swaps.  Same operations in the same order as
// the reference: pivot = first row with the largest |A[j][k]| (strict `<`), rows exchanged (by reference there, by value
// here: same numbers), column scaled by 1/A[k][k] through divisions, rank-1 update with separately rounded product and
// difference, permuted forward substitution, back substitution.  s, d = 4 points x,y as float32; out = h0..h7 (h8 == 1).
HG_HD void solve_projective_regs(const float *s, const float *d, double *out)
{
    double A[8][8];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const double x = s[2 * p], y = s[2 * p + 1], u = d[2 * p], v = d[2 * p + 1];
        A[2 * p][0] = x; A[2 * p][1] = y; A[2 * p][2] = 1; A[2 * p][3] = 0; A[2 * p][4] = 0; A[2 * p][5] = 0;
        A[2 * p][6] = -u * x; A[2 * p][7] = -u * y;                                       // :1322  (-dst) * src
        A[2 * p + 1][0] = 0; A[2 * p + 1][1] = 0; A[2 * p + 1][2] = 0; A[2 * p + 1][3] = x; A[2 * p + 1][4] = y; A[2 * p + 1][5] = 1;
        A[2 * p + 1][6] = -v * x; A[2 * p + 1][7] = -v * y;
    }
    int P[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int pk = k;
        double best = fabs(A[k][k]);
#pragma unroll
        for (int j = k + 1; j < 8; j++) { const double v = fabs(A[j][k]); if (best < v) { best = v; pk = j; } }      // :1713-1718
        P[k] = pk;
#pragma unroll
        for (int i = k + 1; i < 8; i++) {                                                 // :1722-1726 (A[k] <-> A[pk])
            const bool sw = pk == i;
#pragma unroll
            for (int j = 0; j < 8; j++) { const double a = A[k][j], b = A[i][j]; A[k][j] = sw ? b : a; A[i][j] = sw ? a : b; }
        }
        const double akk = A[k][k];
#pragma unroll
        for (int i = k + 1; i < 8; i++) A[i][k] /= akk;                                   // :1730-1732
#pragma unroll
        for (int i = k + 1; i < 8; i++) {
#pragma unroll
            for (int j = k + 1; j < 8; j++) A[i][j] -= A[i][k] * A[k][j];                 // :1734-1742 (the 2-way unrolling there keeps this order)
        }
    }
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = d[i];
#pragma unroll
    for (int i = 0; i < 8; i++) {                                                         // :1673-1685
#pragma unroll
        for (int t = i + 1; t < 8; t++) { const bool sw = P[i] == t; const double a = x[i], b = x[t]; x[i] = sw ? b : a; x[t] = sw ? a : b; }
#pragma unroll
        for (int j = 0; j < i; j++) x[i] -= x[j] * A[i][j];
    }
#pragma unroll
    for (int i = 7; i >= 0; i--) {                                                        // :1687-1694
#pragma unroll
        for (int j = i + 1; j < 8; j++) x[i] -= x[j] * A[i][j];
        x[i] /= A[i][i];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = x[i];
}

// Can every division (m0*x + m1*y + m2) / (m6*x + m7*y + 1), (m3*x + m4*y + m5) / (same) of this frame be done by
// div2_plain (hg_k_geo.hip), i.e. without the scaling / special-value steps of the full IEEE expansion?
//   * matrix entries finite, each 0 or 2^-100 <= |m| <= 2^100; pixel coordinates |x|, |y| < 2^28
//     => numerators are 0 or in [2^-210, 2^130] (a non-zero sum of two such roundings cannot fall below 2^-206);
//   * the denominator, evaluated in the kernel's own operation order, is weakly monotone along x and along y (every
//     rounding is), so over the window it lies between its values at the four corners: same sign at all four and
//     2^-100 <= |den| <= 2^130 there => the same holds at every pixel.
HG_HD bool projective_plain_range(const double *m, int32_t x_off, int32_t y_off, int32_t obj_w, int32_t obj_h)
{
    const double lo = 0x1p-100, hi = 0x1p100;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const double a = fabs(m[k]);
        if (!(a == a) || !(a == 0.0 || (a >= lo && a <= hi))) return false;
    }
    if (obj_w <= 0 || obj_h <= 0) return true;
    const int64_t x0 = x_off, x1 = (int64_t)x_off + obj_w + 255, y0 = y_off, y1 = (int64_t)y_off + obj_h - 1;   // (+255: the ragged last window is computed too)
    const int64_t ax = (x0 < 0 ? -x0 : x0) > (x1 < 0 ? -x1 : x1) ? (x0 < 0 ? -x0 : x0) : (x1 < 0 ? -x1 : x1);
    const int64_t ay = (y0 < 0 ? -y0 : y0) > (y1 < 0 ? -y1 : y1) ? (y0 < 0 ? -y0 : y0) : (y1 < 0 ? -y1 : y1);
    if (ax >= ((int64_t)1 << 28) || ay >= ((int64_t)1 << 28)) return false;
    double dmin = INFINITY, dmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const double x = (double)((c & 1) ? x1 : x0), y = (double)((c & 2) ? y1 : y0);
        const double ad = m[7] * y;
        const double den = ((m[6] * x) + ad) + 1.0;                  // :1402-1403, the kernel's order
        if (!(den == den)) return false;
        dmin = den < dmin ? den : dmin; dmax = den > dmax ? den : dmax;
    }
    if (dmin > 0) return dmin >= lo && dmax <= 0x1p130;
    if (dmax < 0) return -dmax >= lo && -dmin <= 0x1p130;
    return false;
}

// ---------------------------------------------------------------------------------------------- triangle rasteriser

// One edge of defineTriangleLineEquations :1141-1151
struct Seg { double m, b, minY, maxY; };

HG_HD void define_seg(double xa, double ya, double xb, double yb, Seg &s)
{
    if (xb != xa) { s.m = (yb - ya) / (xb - xa); s.b = ya - xa * ((yb - ya) / (xb - xa)); }
    else          { s.m = INFINITY;              s.b = xa; }
    s.minY = js_min2(yb, ya);
    s.maxY = js_max2(yb, ya);
}

// predictXLimits :1172-1197 for integer row y, then the two flat fill() indices of fillTriangle :1124
// ((y - yOffset) * matrix_width + Math.round(x), no x-offset) with TypedArray.fill semantics.
// Returns the half-open cell range [k, fin) actually overwritten (k >= fin: nothing).
HG_HD void span_cells(const Seg *seg, double y, double y_off, double map_w, int64_t len, int64_t &k, int64_t &fin)
{
    double mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (y >= seg[i].minY && y <= seg[i].maxY) {
            double x;
            if (seg[i].m == INFINITY) x = seg[i].b;
            else if (seg[i].m == 0.0) continue;           // horizontal edge (also -0): skipped
            else x = (y - seg[i].b) / seg[i].m;
            if (x < mn) mn = x;
            if (x > mx) mx = x;
        }
    }
    const double base = (y - y_off) * map_w;
    k = js_fill_index(base + js_round(mn), len);
    fin = js_fill_index(base + js_round(mx), len);
}

// Row range of fillTriangle :1113-1120: y from ~~min(y) while y < ceil(max(y)).  y_end is clamped to int range;
// NaN (no rows) gives y_end = y_min.
HG_HD void tri_rows(double y0, double y1, double y2, int32_t &y_min, int32_t &y_end)
{
    const double mn = js_min2(js_min2(y0, y1), y2), mx = js_max2(js_max2(y0, y1), y2);
    y_min = js_to_int32(mn);
    const double top = ceil(mx);
    if (top != top) { y_end = y_min; return; }
    y_end = top > 1073741824.0 ? 1073741824 : (top < -1073741824.0 ? -1073741824 : (int32_t)top);
}

// The rows of that loop that can have a span at all: predictXLimits :1179 only takes edges with minY <= y <= maxY, and every edge's
// minY is at least the triangle's, so the loop's first row trunc(minY) is a no-op whenever minY is not an integer (mn = +Inf,
// mx = -Inf -> fill(idx, len, 0)).  Consumers that enumerate candidate rows (the self-span prologues) start at ceil(minY).
HG_HD void tri_rows_tight(double y0, double y1, double y2, int32_t &y_min, int32_t &y_end)
{
    tri_rows(y0, y1, y2, y_min, y_end);
    const double mn = js_min2(js_min2(y0, y1), y2);
    if (fabs(mn) < 1073741824.0) { const int32_t c = (int32_t)ceil(mn); if (c > y_min) y_min = c; }
}

// Rows of fillTriangle's loop that can overwrite a cell at all.  A row y writes cells (y - yOff) * W + round(x) (after
// TypedArray.fill's index rules: indices <= -len and >= len write nothing) with x between the triangle's vertex x's, and
// the library only accepts coordinates up to kMaxCoord in magnitude (hg_piecewise_set_frames / _set_mesh; NaN vertices
// produce no cells).  So every row outside (yOff - (len + kMaxCoord + 2) / W, yOff + (len + kMaxCoord + 2) / W) is a no-op
// and the kernels skip it: the loop length is bounded by the window, not by how far away a vertex lies.
constexpr double kMaxCoord = 16777216.0;        // 2^24

HG_HD void clamp_rows(int64_t &y_first, int64_t &y_end, int32_t y_off, int32_t map_w, int64_t len)
{
    if (map_w <= 0) { y_end = y_first; return; }
    const double reach = ((double)len + kMaxCoord + 2.0) / (double)map_w + 2.0;
    const int64_t lo = (int64_t)floor((double)y_off - reach), hi = (int64_t)ceil((double)y_off + reach);
    if (y_first < lo) y_first = lo;
    if (y_end > hi) y_end = hi;
}

// ---------------------------------------------------------------------------------------------- point transforms

// applyAffineTransformToPoint :1382-1385 (m = 6 doubles holding f32 values)
HG_HD void apply_affine(const double *m, double x, double y, double &ox, double &oy)
{
    ox = (m[0] * x) + (m[2] * y) + m[4];
    oy = (m[1] * x) + (m[3] * y) + m[5];
}
// When may (m0*x) + (m2*y) + m4 of :1383-1384 be computed as ONE fma(m0, x, c) with c = (m2*y) + m4 ?
// The reference rounds twice: RN(RN(m0 x + m2 y) + m4) (both products are exact in fp64: an f32 value times an integer below 2^28).
// If (i) m0 x + m2 y is exactly representable, the reference's result is RN(m0 x + m2 y + m4); if (ii) m2 y + m4 is exactly representable,
// fma(m0, x, c) is RN(m0 x + m2 y + m4) as well: the same bits.  Both sums are integer multiples of 2^g (g = the smallest unit in the last
// place of the f32 operands involved; x, y are integers) and smaller in magnitude than 2^(top + 1): representable whenever
// top + 1 <= 53 + g.  lx, ly: |x| <= 2^lx, |y| <= 2^ly for every pixel of the frame (bit lengths of the window's extreme coordinates).
// Conservative (never says yes wrongly); non-finite entries say no.  Zero entries contribute nothing to either sum.
HG_HD bool affine_row_fusable(float m0, float m2, float m4, int lx, int ly)
{
    union { float f; uint32_t u; } a, b, d;
    a.f = m0; b.f = m2; d.f = m4;
    const int ea = (int)((a.u >> 23) & 0xff), eb = (int)((b.u >> 23) & 0xff), ed = (int)((d.u >> 23) & 0xff);
    if (ea == 255 || eb == 255 || ed == 255) return false;
    const bool za = (a.u << 1) == 0, zb = (b.u << 1) == 0, zd = (d.u << 1) == 0;
    // f32 v = k * 2^(E - 150), |k| < 2^24, |v| < 2^(E - 126), with E = max(biased exponent, 1)
    const int Ea = ea ? ea : 1, Eb = eb ? eb : 1, Ed = ed ? ed : 1;
    const int big = 1 << 20;
    const int ua = za ? big : Ea - 150, ub = zb ? big : Eb - 150, ud = zd ? big : Ed - 150;
    const int ta = za ? -big : Ea - 126 + lx, tb = zb ? -big : Eb - 126 + ly, td = zd ? -big : Ed - 126;
    const int g1 = ua < ub ? ua : ub, top1 = ta > tb ? ta : tb;       // (i)  m0 x + m2 y
    const int g2 = ub < ud ? ub : ud, top2 = tb > td ? tb : td;       // (ii) m2 y + m4
    return top1 + 1 <= 53 + g1 && top2 + 1 <= 53 + g2;               // (an all-zero sum: -big + 1 <= 53 + big)
}
// number of bits of the largest |coordinate| of a window [off, off + n): |x| <= 2^bits
HG_HD int coord_bits(int off, int n)
{
    int64_t a = off < 0 ? -(int64_t)off : off, b = (int64_t)off + n;
    if (b < 0) b = -b;
    uint64_t m = (uint64_t)(a > b ? a : b);
    int bits = 0;
    while (m) { bits++; m >>= 1; }
    return bits;
}
// both rows of an inverse affine matrix [m0 m1 m2 m3 m4 m5] (sx = m0 x + m2 y + m4, sy = m1 x + m3 y + m5)
HG_HD bool affine_fusable(const float *inv, int lx, int ly)
{
    return affine_row_fusable(inv[0], inv[2], inv[4], lx, ly) && affine_row_fusable(inv[1], inv[3], inv[5], lx, ly);
}

// applyProjectiveTransformToPoint :1401-1404 (the denominator is evaluated twice in JS; same value both times)
HG_HD void apply_projective(const double *m, double x, double y, double &ox, double &oy)
{
    const double den = m[6] * x + m[7] * y + 1;
    ox = (m[0] * x + m[1] * y + m[2]) / den;
    oy = (m[3] * x + m[4] * y + m[5]) / den;
}

} // namespace hg
