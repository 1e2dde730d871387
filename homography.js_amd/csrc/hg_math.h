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
// applyProjectiveTransformToPoint :1401-1404 (the denominator is evaluated twice in JS; same value both times)
HG_HD void apply_projective(const double *m, double x, double y, double &ox, double &oy)
{
    const double den = m[6] * x + m[7] * y + 1;
    ox = (m[0] * x + m[1] * y + m[2]) / den;
    oy = (m[3] * x + m[4] * y + m[5]) / den;
}

} // namespace hg
