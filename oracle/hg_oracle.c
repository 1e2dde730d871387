/*
 * hg_oracle.c -- see hg_oracle.h.  TEST INFRASTRUCTURE ONLY (checker + cpu_baseline); never on the product path.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared  (oracle/Makefile)
 * Citations are file:line into /root/reference/Homography.js (v1.8.0).
 */
#include "hg_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ JS number helpers */

double hgo_js_round(double x)
{
    /* ECMA-262 Math.round: ties toward +Infinity; NaN, +-Inf, and |x| >= 2^52 pass through. */
    if (!(fabs(x) < 4503599627370496.0)) return x;
    double r = floor(x);
    if (x - r >= 0.5) r += 1.0;            /* x - floor(x) is exact */
    return r;
}

static int32_t js_to_int32(double x)
{
    /* ECMA-262 ToInt32 (what `~~v` and `v << 2` apply to their operand). */
    if (!isfinite(x)) return 0;
    double t = trunc(x);
    double m = fmod(t, 4294967296.0);
    if (m < 0) m += 4294967296.0;
    uint32_t u = (uint32_t)m;
    return (int32_t)u;
}

static int32_t js_shl2(double x) { return (int32_t)((uint32_t)js_to_int32(x) << 2); }   /* `x << 2` */

static double js_min3(double a, double b, double c)
{
    if (isnan(a) || isnan(b) || isnan(c)) return NAN;
    double m = a < b ? a : b;
    return m < c ? m : c;
}
static double js_max3(double a, double b, double c)
{
    if (isnan(a) || isnan(b) || isnan(c)) return NAN;
    double m = a > b ? a : b;
    return m > c ? m : c;
}
static double js_min2(double a, double b) { if (isnan(a) || isnan(b)) return NAN; return a < b ? a : b; }
static double js_max2(double a, double b) { if (isnan(a) || isnan(b)) return NAN; return a > b ? a : b; }
static double js_min4(double a, double b, double c, double d) { return js_min2(js_min2(a, b), js_min2(c, d)); }
static double js_max4(double a, double b, double c, double d) { return js_max2(js_max2(a, b), js_max2(c, d)); }

/* TypedArray.prototype.fill relative index: ToIntegerOrInfinity, negative counts from the end, clamp to [0,len]. */
static int64_t js_fill_index(double v, int64_t len)
{
    if (isnan(v)) v = 0.0;
    if (v == -INFINITY) return 0;
    if (v == INFINITY) return len;
    v = trunc(v);
    if (v < 0) { double k = (double)len + v; return k < 0 ? 0 : (int64_t)k; }
    return v > (double)len ? len : (int64_t)v;
}

/* ------------------------------------------------------------------ transform solves */

void hgo_affine_from_triangles(const float s[6], const float d[6], float out[6])
{
    /* :1269-1284 translate both triangles by their third vertex */
    const double srcE = s[4], srcF = s[5];
    const double srcA = s[0] - srcE, srcB = s[1] - srcF, srcC = s[2] - srcE, srcD = s[3] - srcF;
    const double dstE = d[4], dstF = d[5];
    const double dstA = d[0] - dstE, dstB = d[1] - dstF, dstC = d[2] - dstE, dstD = d[3] - dstF;
    /* :1287-1294 inverse of the source 2x3 (note the `x / -den` pattern) */
    const double den = srcA * srcD - srcB * srcC;
    const double iA = srcD / den;
    const double iB = srcB / -den;
    const double iC = srcC / -den;
    const double iD = srcA / den;
    const double iE = (srcD * srcE - srcC * srcF) / -den;
    const double iF = (srcB * srcE - srcA * srcF) / den;
    /* :1297-1304 dst * inv(src), stored through a Float32Array */
    out[0] = (float)((dstA * iA) + (dstC * iB));
    out[1] = (float)((dstB * iA) + (dstD * iB));
    out[2] = (float)((dstA * iC) + (dstC * iD));
    out[3] = (float)((dstB * iC) + (dstD * iD));
    out[4] = (float)((dstA * iE) + (dstC * iF) + dstE);
    out[5] = (float)((dstB * iE) + (dstD * iF) + dstF);
}

void hgo_inverse_affine(const float m[6], float out[6])
{
    /* :1346-1362 */
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5];
    const double den = a * d - b * c;
    out[0] = (float)(d / den);
    out[1] = (float)(b / -den);
    out[2] = (float)(c / -den);
    out[3] = (float)(a / den);
    out[4] = (float)((d * e - c * f) / -den);
    out[5] = (float)((b * e - a * f) / den);
}

void hgo_projective_from_squares(const float s[8], const float d[8], double out[8])
{
    /* :1322-1329 the 8x8 DLT system, rows [x y 1 0 0 0 -u*x -u*y] / [0 0 0 x y 1 -v*x -v*y] */
    double Abuf[8][8];
    double *A[8];
    for (int p = 0; p < 4; p++) {
        const double x = s[2 * p], y = s[2 * p + 1], u = d[2 * p], v = d[2 * p + 1];
        double *r0 = Abuf[2 * p], *r1 = Abuf[2 * p + 1];
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * x; r0[7] = -u * y;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -v * x; r1[7] = -v * y;
    }
    for (int i = 0; i < 8; i++) A[i] = Abuf[i];
    /* LU :1697-1750  Doolittle, partial pivoting with strict '<' (first maximum wins), rows swapped by pointer */
    int P[8];
    const int n = 8;
    for (int k = 0; k < n; k++) {
        int Pk = k;
        double *Ak = A[k];
        double max = fabs(Ak[k]);
        for (int j = k + 1; j < n; j++) {
            const double absAjk = fabs(A[j][k]);
            if (max < absAjk) { max = absAjk; Pk = j; }
        }
        P[k] = Pk;
        if (Pk != k) { A[k] = A[Pk]; A[Pk] = Ak; Ak = A[k]; }
        const double Akk = Ak[k];
        for (int i = k + 1; i < n; i++) A[i][k] /= Akk;
        for (int i = k + 1; i < n; i++) {
            double *Ai = A[i];
            for (int j = k + 1; j < n; j++) Ai[j] -= Ai[k] * Ak[j];      /* :1734-1742 (2-way unrolled there; same per-element order) */
        }
    }
    /* LUsolve :1664-1695 */
    double x[8];
    for (int i = 0; i < n; i++) x[i] = d[i];
    for (int i = 0; i < n; i++) {
        const int Pi = P[i];
        if (Pi != i) { const double t = x[i]; x[i] = x[Pi]; x[Pi] = t; }
        const double *LUi = A[i];
        for (int j = 0; j < i; j++) x[i] -= x[j] * LUi[j];
    }
    for (int i = n - 1; i >= 0; i--) {
        const double *LUi = A[i];
        for (int j = i + 1; j < n; j++) x[i] -= x[j] * LUi[j];
        x[i] /= LUi[i];
    }
    for (int i = 0; i < n; i++) out[i] = x[i];
}

/* ------------------------------------------------------------------ triangle-index map */

typedef struct { double m, b, minY, maxY; } hgo_seg;

static void define_segment(double xa, double ya, double xb, double yb, hgo_seg *s)
{
    /* :1145-1149  m = (yb-ya)/(xb-xa) or Infinity; b recomputes the slope expression; vertical => b = xa */
    if (xb != xa) { s->m = (yb - ya) / (xb - xa); s->b = ya - xa * ((yb - ya) / (xb - xa)); }
    else          { s->m = INFINITY;              s->b = xa; }
    s->minY = js_min2(yb, ya);
    s->maxY = js_max2(yb, ya);
}

void hgo_fill_triangle(const float t[6], double idx, double matrix_width, double y_offset, int16_t *map, int64_t len)
{
    const double x0 = t[0], y0 = t[1], x1 = t[2], y1 = t[3], x2 = t[4], y2 = t[5];
    const double minY = (double)js_to_int32(js_min3(y0, y1, y2));     /* :1113  ~~Math.min(...) */
    const double maxY = ceil(js_max3(y0, y1, y2));                    /* :1115 */
    hgo_seg seg[3];
    define_segment(x0, y0, x1, y1, &seg[0]);                          /* p0->p1 */
    define_segment(x0, y0, x2, y2, &seg[1]);                          /* p0->p2 */
    define_segment(x1, y1, x2, y2, &seg[2]);                          /* p1->p2 */
    const int16_t v = (int16_t)(uint16_t)js_to_int32(idx);            /* Int16Array element conversion (mod 2^16) */
    for (double y = minY; y < maxY; y += 1.0) {                       /* :1120 (false when maxY is NaN) */
        double mn = INFINITY, mx = -INFINITY;                         /* :1175-1176 */
        for (int i = 0; i < 3; i++) {
            if (y >= seg[i].minY && y <= seg[i].maxY) {               /* :1179 */
                double x;
                if (seg[i].m == INFINITY) x = seg[i].b;               /* :1181 */
                else if (seg[i].m == 0.0) continue;                   /* :1184 (also -0) */
                else x = (y - seg[i].b) / seg[i].m;                   /* :1188 */
                if (x < mn) mn = x;
                if (x > mx) mx = x;
            }
        }
        const double start = (y - y_offset) * matrix_width + hgo_js_round(mn);    /* :1124 flat index, no x-offset */
        const double end   = (y - y_offset) * matrix_width + hgo_js_round(mx);
        const int64_t k = js_fill_index(start, len), fin = js_fill_index(end, len);
        for (int64_t c = k; c < fin; c++) map[c] = v;
    }
}

void hgo_build_tri_map(const float *pts, const uint32_t *tris, int n_tris, double matrix_width, double y_offset,
                       int16_t *map, int64_t len)
{
    for (int64_t c = 0; c < len; c++) map[c] = -1;                    /* :850 / :822 */
    for (int i = 0; i < n_tris; i++) {
        float t[6];                                                   /* :854-856 gather through Float32Array(6) */
        for (int k = 0; k < 3; k++) { t[2 * k] = pts[2 * (size_t)tris[3 * i + k]]; t[2 * k + 1] = pts[2 * (size_t)tris[3 * i + k] + 1]; }
        hgo_fill_triangle(t, (double)i, matrix_width, y_offset, map, len);   /* :857 idx = i/3 */
    }
}

void hgo_piecewise_matrices(const float *sp, const float *dp, const uint32_t *tris, int n_tris, float *fwd)
{
    for (int i = 0; i < n_tris; i++) {                                /* :791-802 */
        float s[6], d[6];
        for (int k = 0; k < 3; k++) {
            const size_t v = tris[3 * i + k];
            s[2 * k] = sp[2 * v]; s[2 * k + 1] = sp[2 * v + 1];
            d[2 * k] = dp[2 * v]; d[2 * k + 1] = dp[2 * v + 1];
        }
        hgo_affine_from_triangles(s, d, fwd + 6 * (size_t)i);
    }
}

/* ------------------------------------------------------------------ geometry */

static void apply_affine(const double *m, double x, double y, double *ox, double *oy)
{
    *ox = (m[0] * x) + (m[2] * y) + m[4];                             /* :1383 */
    *oy = (m[1] * x) + (m[3] * y) + m[5];                             /* :1384 */
}
static void apply_projective(const double *m, double x, double y, double *ox, double *oy)
{
    *ox = (m[0] * x + m[1] * y + m[2]) / (m[6] * x + m[7] * y + 1);   /* :1402 */
    *oy = (m[3] * x + m[4] * y + m[5]) / (m[6] * x + m[7] * y + 1);   /* :1403 */
}

void hgo_transform_limits(int kind, const double *m, double width, double height, double out[4])
{
    double p00[2], p10[2], p01[2], p11[2];                            /* :1506-1517 */
    void (*T)(const double *, double, double, double *, double *) = kind == 0 ? apply_affine : apply_projective;
    T(m, 0, 0, &p00[0], &p00[1]);
    T(m, 0, height, &p10[0], &p10[1]);
    T(m, width, 0, &p01[0], &p01[1]);
    T(m, width, height, &p11[0], &p11[1]);
    const double xo = js_min4(p00[0], p10[0], p01[0], p11[0]);       /* :1521-1524 */
    const double yo = js_min4(p00[1], p01[1], p10[1], p11[1]);
    const double ow = js_max4(p01[0], p11[0], p00[0], p10[0]) - xo;
    const double oh = js_max4(p10[1], p11[1], p00[1], p01[1]) - yo;
    out[0] = hgo_js_round(xo); out[1] = hgo_js_round(yo); out[2] = hgo_js_round(ow); out[3] = hgo_js_round(oh);   /* :1525 */
}

void hgo_minmax_xy(const float *p, int n, double out[4])
{
    double maxX = -INFINITY, maxY = -INFINITY, minX = INFINITY, minY = INFINITY;    /* :1560-1563 */
    for (int i = 0; i < n; i++) {
        const double e = p[i];
        if ((i % 2) == 0) { if (e > maxX) maxX = e; if (e < minX) minX = e; }
        else              { if (e > maxY) maxY = e; if (e < minY) minY = e; }
    }
    out[0] = hgo_js_round(minX); out[1] = hgo_js_round(minY); out[2] = hgo_js_round(maxX); out[3] = hgo_js_round(maxY);   /* :1585 */
}

/* ------------------------------------------------------------------ warps */

/* image[idx..idx+3] -> out[o..o+3]; reads outside the source array give `undefined` => 0 in a Uint8ClampedArray */
static inline void copy_px(const uint8_t *image, int64_t n_src, double src_idx, uint8_t *out, int64_t o)
{
    if (src_idx >= 0 && src_idx + 3 < (double)n_src) {
        const int64_t s = (int64_t)src_idx;
        out[o] = image[s]; out[o + 1] = image[s + 1]; out[o + 2] = image[s + 2]; out[o + 3] = image[s + 3];
    } else {
        /* flat source indices are always multiples of 4, so the 4 channels are in or out together */
        out[o] = out[o + 1] = out[o + 2] = out[o + 3] = 0;
    }
}

void hgo_warp_inverse_geometric(int kind, const double *m, const uint8_t *image, int W, int H,
                                int xOff, int yOff, int objW, int objH, uint8_t *out)
{
    const double srcRow = (double)((int64_t)W << 2);
    const int64_t dstRow = (int64_t)objW << 2, n_src = (int64_t)W * H * 4;
    if (objW <= 0 || objH <= 0) return;
    memset(out, 0, (size_t)dstRow * (size_t)objH);                    /* :991 */
    for (int y = yOff; y < objH + yOff; y++) {                        /* :997 */
        for (int x = xOff; x < objW + xOff; x++) {                    /* :998 */
            double sx, sy;
            if (kind == 0) apply_affine(m, x, y, &sx, &sy); else apply_projective(m, x, y, &sx, &sy);   /* :999 */
            if (sx >= 0 && sx < W && sy >= 0 && sy < H) {             /* :1001 (test on the unrounded coordinate) */
                const int64_t idx = (int64_t)(y - yOff) * dstRow + ((int64_t)(x - xOff) << 2);           /* :1003 */
                const double srcIdx = (hgo_js_round(sy) * srcRow) + (double)js_shl2(hgo_js_round(sx));  /* :1005 */
                copy_px(image, n_src, srcIdx, out, idx);              /* :1006-1007 */
            }
        }
    }
}

void hgo_warp_inverse_piecewise_loop(const int16_t *map, const float *inv, int n_tris, const uint8_t *image, int W, int H,
                                     int minSrcX, int minSrcY, int xOff, int yOff, int objW, int objH, uint8_t *out)
{
    (void)n_tris;
    const double srcRow = (double)((int64_t)W << 2);
    const int64_t dstRow = (int64_t)objW << 2, n_src = (int64_t)W * H * 4;
    if (objW <= 0 || objH <= 0) return;
    memset(out, 0, (size_t)dstRow * (size_t)objH);                    /* :1040 */
    for (int y = yOff; y < objH + yOff; y++) {                        /* :1042 */
        for (int x = xOff; x < objW + xOff; x++) {                    /* :1043 */
            const int16_t t = map[(int64_t)(y - yOff) * objW + (x - xOff)];      /* :1044 */
            if (t >= 0) {                                             /* :1045 */
                const float *mf = inv + 6 * (size_t)t;
                const double m[6] = { mf[0], mf[1], mf[2], mf[3], mf[4], mf[5] };
                double sx, sy;
                apply_affine(m, x, y, &sx, &sy);                      /* :1046 */
                if (sx >= minSrcX && sx < W + minSrcX && sy >= minSrcY && sy < H + minSrcY) {   /* :1047 */
                    sx = hgo_js_round(sx); sy = hgo_js_round(sy);     /* :1048 */
                    const double srcIdx = (sy * srcRow) + (double)js_shl2(sx);                   /* :1049 */
                    const int64_t dstIdx = (int64_t)(y - yOff) * dstRow + ((int64_t)(x - xOff) << 2);   /* :1050 */
                    copy_px(image, n_src, srcIdx, out, dstIdx);       /* :1051-1052 */
                }
            }
        }
    }
}

void hgo_warp_inverse_piecewise(const float *sp, const float *dp, const uint32_t *tris, int n_tris,
                                const uint8_t *image, int W, int H, int minSrcX, int minSrcY,
                                int xOff, int yOff, int objW, int objH,
                                uint8_t *out, int16_t *map_out, float *fwd_out, float *inv_out)
{
    const int64_t len = (objW > 0 && objH > 0) ? (int64_t)objW * objH : 0;
    float *fwd = fwd_out ? fwd_out : (float *)malloc(sizeof(float) * 6 * (size_t)(n_tris > 0 ? n_tris : 1));
    float *inv = inv_out ? inv_out : (float *)malloc(sizeof(float) * 6 * (size_t)(n_tris > 0 ? n_tris : 1));
    int16_t *map = map_out ? map_out : (int16_t *)malloc(sizeof(int16_t) * (size_t)(len > 0 ? len : 1));
    hgo_piecewise_matrices(sp, dp, tris, n_tris, fwd);                               /* setDestinyPoints -> :769 */
    hgo_build_tri_map(dp, tris, n_tris, (double)objW, (double)yOff, map, len);       /* :1033 */
    for (int i = 0; i < n_tris; i++) hgo_inverse_affine(fwd + 6 * (size_t)i, inv + 6 * (size_t)i);   /* :1036-1038 */
    hgo_warp_inverse_piecewise_loop(map, inv, n_tris, image, W, H, minSrcX, minSrcY, xOff, yOff, objW, objH, out);
    if (!fwd_out) free(fwd);
    if (!inv_out) free(inv);
    if (!map_out) free(map);
}

/* typed-array store at a computed (double) index: out-of-range / non-integer / NaN indices are silently ignored */
static inline void store_px(uint8_t *out, int64_t n_dst, double new_idx, const uint8_t *px)
{
    if (new_idx >= 0 && new_idx + 3 < (double)n_dst) {
        const int64_t o = (int64_t)new_idx;
        out[o] = px[0]; out[o + 1] = px[1]; out[o + 2] = px[2]; out[o + 3] = px[3];
    }
}

void hgo_warp_forward_geometric(int kind, const double *m, const uint8_t *image, int W, int H,
                                int xOff, int yOff, int objW, int objH, uint8_t *out)
{
    const int64_t srcRow = (int64_t)W << 2, n_dst = (objW > 0 && objH > 0) ? ((int64_t)objW << 2) * objH : 0;
    const double dstRow = (double)((int64_t)objW << 2);
    if (n_dst > 0) memset(out, 0, (size_t)n_dst);                     /* :916 */
    for (int y = 0; y < H; y++) {                                     /* :919 */
        for (int x = 0; x < W; x++) {
            const int64_t idx = y * srcRow + ((int64_t)x << 2);       /* :922 */
            double nx, ny;
            if (kind == 0) apply_affine(m, x, y, &nx, &ny); else apply_projective(m, x, y, &nx, &ny);   /* :923 */
            nx = hgo_js_round(nx - xOff); ny = hgo_js_round(ny - yOff);                                 /* :924 */
            const double newIdx = (ny * dstRow) + (double)js_shl2(nx);                                  /* :926 */
            store_px(out, n_dst, newIdx, image + idx);                /* :927-928 */
        }
    }
}

void hgo_warp_forward_piecewise(const int16_t *fmap, const float *fwd, int n_tris, const uint8_t *image, int W, int H,
                                int minSrcX, int minSrcY, int maxSrcX, int maxSrcY,
                                int xOff, int yOff, int objW, int objH, uint8_t *out)
{
    (void)n_tris;
    const int64_t srcRow = (int64_t)W << 2, n_dst = (objW > 0 && objH > 0) ? ((int64_t)objW << 2) * objH : 0;
    const int64_t n_src = (int64_t)W * H * 4;
    const double dstRow = (double)((int64_t)objW << 2);
    const int64_t mapW = (int64_t)maxSrcX - minSrcX;                  /* :951 */
    if (n_dst > 0) memset(out, 0, (size_t)n_dst);                     /* :953 */
    for (int y = minSrcY; y < maxSrcY; y++) {                         /* :955 */
        for (int x = minSrcX; x < maxSrcX; x++) {
            const int16_t t = fmap[(int64_t)(y - minSrcY) * mapW + (x - minSrcX)];   /* :957 */
            if (t > -1) {
                const int64_t idx = y * srcRow + ((int64_t)x << 2);  /* :960 */
                const float *mf = fwd + 6 * (size_t)t;
                const double m[6] = { mf[0], mf[1], mf[2], mf[3], mf[4], mf[5] };
                double nx, ny;
                apply_affine(m, x, y, &nx, &ny);                      /* :961 */
                nx = hgo_js_round(nx - xOff); ny = hgo_js_round(ny - yOff);          /* :962 */
                const double newIdx = (ny * dstRow) + (double)js_shl2(nx);           /* :964 */
                uint8_t px[4] = { 0, 0, 0, 0 };                       /* source reads outside the array are `undefined` => 0 */
                if (idx >= 0 && idx + 3 < n_src) memcpy(px, image + idx, 4);
                store_px(out, n_dst, newIdx, px);                     /* :966-967 */
            }
        }
    }
}

void hgo_lcg_image(uint8_t *data, size_t n, uint32_t seed)
{
    uint32_t s = seed;
    for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; data[i] = (uint8_t)(s >> 24); }
}
