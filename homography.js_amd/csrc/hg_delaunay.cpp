// hg_delaunay.cpp -- host Delaunay triangulator behind hg_triangulate (include/hgwarp.h).
//
// Stands where the reference calls `new Delaunator(points).triangles` (Homography.js:1216-1218 <- :262, :742).  The
// reference's dependency, delaunator 5.0.0 (+ robust-predicates 3.0.1: package.json:11, package-lock.json:17-37), is not
// vendored, and the piecewise result depends on the triangle LIST (diagonal of every cocircular quad, order of the
// triangles: the largest id wins where spans overlap).  This file therefore restates delaunator 5's published algorithm --
// seed triangle, points sorted by distance from the seed circumcentre with its quicksort, advancing convex hull with the
// pseudo-angle hash, edge-flip legalisation with a fixed 512-entry stack, half-edge links -- in the same double arithmetic
// (compiled with -ffp-contract=off), with the orientation test as an exact-sign predicate (only its sign is used, as there).
// js/delaunay.mjs is the same code in JavaScript; both return the identical list (tests/js/test_host.mjs).
// None of the reference's tests pins delaunator's output, so this remains "triangulation parity unpinned" (SURVEY.md §8c).
//
// ATTRIBUTION.  This is a restatement (written from the published algorithm, not a copy of the source, which is absent from this
// build environment) of third-party work; see THIRD_PARTY_NOTICES.md at the repository root:
//   delaunator 5.0.0          Copyright (c) 2017, Mapbox -- ISC License                (https://github.com/mapbox/delaunator)
//   robust-predicates 3.0.1   Vladimir Agafonkin -- The Unlicense (public domain); its orient2d is a port of Jonathan R. Shewchuk's
//                             public-domain "Adaptive Precision Floating-Point Arithmetic and Fast Robust Geometric Predicates"
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/hgwarp.h"

namespace {

// ---- exact sign of (ay - cy) * (bx - cx) - (ax - cx) * (by - cy): robust-predicates' fast path and error bound, then
// Shewchuk's exact expansion arithmetic (TwoDiff tails, TwoProduct by Veltkamp splitting, Grow-Expansion)
struct Expansion {
    double e[40]; int n = 0;
    void grow(double b) {
        double Q = b;
        for (int i = 0; i < n; i++) { const double x = Q + e[i], bv = x - Q, av = x - bv; e[i] = (Q - av) + (e[i] - bv); Q = x; }
        e[n++] = Q;
    }
    void add_product(double a, double b, double sign) {
        const double x = a * b;
        double c = 134217729.0 * a; const double ahi = c - (c - a), alo = a - ahi;
        c = 134217729.0 * b; const double bhi = c - (c - b), blo = b - bhi;
        const double y = alo * blo - (((x - ahi * bhi) - alo * bhi) - ahi * blo);
        grow(sign * y); grow(sign * x);
    }
};
inline double two_diff_tail(double a, double b, double x) { const double bv = a - x, av = x + bv; return (a - av) + (bv - b); }

double orient2d(double ax, double ay, double bx, double by, double cx, double cy)
{
    const double detleft = (ay - cy) * (bx - cx), detright = (ax - cx) * (by - cy), det = detleft - detright;
    const double detsum = std::fabs(detleft + detright);
    const double bound = (3 + 16 * 1.1102230246251565e-16) * 1.1102230246251565e-16;
    if (std::fabs(det) >= bound * detsum) return det;
    if (!(std::fabs(ax) < 1e150 && std::fabs(ay) < 1e150 && std::fabs(bx) < 1e150 && std::fabs(by) < 1e150 && std::fabs(cx) < 1e150 && std::fabs(cy) < 1e150)) return det;
    const double p = ay - cy, q = bx - cx, r = ax - cx, s = by - cy;
    const double pt = two_diff_tail(ay, cy, p), qt = two_diff_tail(bx, cx, q), rt = two_diff_tail(ax, cx, r), st = two_diff_tail(by, cy, s);
    Expansion E;
    E.add_product(pt, qt, 1); E.add_product(rt, st, -1);
    E.add_product(p, qt, 1); E.add_product(pt, q, 1); E.add_product(r, st, -1); E.add_product(rt, s, -1);
    E.add_product(p, q, 1); E.add_product(r, s, -1);
    for (int i = E.n - 1; i >= 0; i--) if (E.e[i] != 0) return E.e[i] > 0 ? 1.0 : -1.0;
    return 0.0;
}

inline double pseudo_angle(double dx, double dy) { const double p = dx / (std::fabs(dx) + std::fabs(dy)); return (dy > 0 ? 3 - p : 1 + p) / 4; }
inline double dist2(double ax, double ay, double bx, double by) { const double dx = ax - bx, dy = ay - by; return dx * dx + dy * dy; }
inline bool in_circle(double ax, double ay, double bx, double by, double cx, double cy, double px, double py)
{
    const double dx = ax - px, dy = ay - py, ex = bx - px, ey = by - py, fx = cx - px, fy = cy - py;
    const double ap = dx * dx + dy * dy, bp = ex * ex + ey * ey, cp = fx * fx + fy * fy;
    return dx * (ey * cp - bp * fy) - dy * (ex * cp - bp * fx) + ap * (ex * fy - ey * fx) < 0;
}
inline double circumradius(double ax, double ay, double bx, double by, double cx, double cy)
{
    const double dx = bx - ax, dy = by - ay, ex = cx - ax, ey = cy - ay;
    const double bl = dx * dx + dy * dy, cl = ex * ex + ey * ey, d = 0.5 / (dx * ey - dy * ex);
    const double x = (ey * bl - dy * cl) * d, y = (dx * cl - ex * bl) * d;
    return x * x + y * y;
}

void quicksort(std::vector<uint32_t> &ids, const std::vector<double> &dists, int left, int right)
{
    if (right - left <= 20) {
        for (int i = left + 1; i <= right; i++) {
            const uint32_t temp = ids[i]; const double td = dists[temp];
            int j = i - 1;
            while (j >= left && dists[ids[j]] > td) { ids[j + 1] = ids[j]; j--; }
            ids[j + 1] = temp;
        }
    } else {
        const int median = (left + right) >> 1;
        int i = left + 1, j = right;
        std::swap(ids[median], ids[i]);
        if (dists[ids[left]] > dists[ids[right]]) std::swap(ids[left], ids[right]);
        if (dists[ids[i]] > dists[ids[right]]) std::swap(ids[i], ids[right]);
        if (dists[ids[left]] > dists[ids[i]]) std::swap(ids[left], ids[i]);
        const uint32_t temp = ids[i]; const double td = dists[temp];
        while (true) {
            do i++; while (dists[ids[i]] < td);
            do j--; while (dists[ids[j]] > td);
            if (j < i) break;
            std::swap(ids[i], ids[j]);
        }
        ids[left + 1] = ids[j];
        ids[j] = temp;
        if (right - i + 1 >= j - left) { quicksort(ids, dists, i, right); quicksort(ids, dists, left, j - 1); }
        else { quicksort(ids, dists, left, j - 1); quicksort(ids, dists, i, right); }
    }
}

struct Sweep {
    const std::vector<double> &C;               // x0, y0, x1, y1, ...
    std::vector<uint32_t> triangles, hullPrev, hullNext, hullTri;
    std::vector<int32_t> halfedges, hullHash;
    int hashSize = 0, trianglesLen = 0;
    uint32_t hullStart = 0;
    double ccx = 0, ccy = 0;
    uint32_t stack[512];

    explicit Sweep(const std::vector<double> &c) : C(c) {}
    void link(int a, int b) { halfedges[a] = b; if (b != -1) halfedges[b] = a; }
    int add_triangle(uint32_t i0, uint32_t i1, uint32_t i2, int a, int b, int c)
    {
        const int t = trianglesLen;
        triangles[t] = i0; triangles[t + 1] = i1; triangles[t + 2] = i2;
        link(t, a); link(t + 1, b); link(t + 2, c);
        trianglesLen += 3;
        return t;
    }
    int hash_key(double x, double y) const
    {
        const double k = std::fmod(std::floor(pseudo_angle(x - ccx, y - ccy) * hashSize), (double)hashSize);
        return k == k ? (int)k : 0;              // (a point exactly on the circumcentre has no angle)
    }
    int legalize(int a)
    {
        int i = 0, ar = 0;
        while (true) {
            const int b = halfedges[a];
            const int a0 = a - a % 3;
            ar = a0 + (a + 2) % 3;
            if (b == -1) {                        // convex hull edge
                if (i == 0) break;
                a = (int)stack[--i];
                continue;
            }
            const int b0 = b - b % 3, al = a0 + (a + 1) % 3, bl = b0 + (b + 2) % 3;
            const uint32_t p0 = triangles[ar], pr = triangles[a], pl = triangles[al], p1 = triangles[bl];
            const bool illegal = in_circle(C[2 * p0], C[2 * p0 + 1], C[2 * pr], C[2 * pr + 1], C[2 * pl], C[2 * pl + 1], C[2 * p1], C[2 * p1 + 1]);
            if (illegal) {
                triangles[a] = p1;
                triangles[b] = p0;
                const int hbl = halfedges[bl];
                if (hbl == -1) {                  // edge swapped on the other side of the hull (rare): fix the hull's triangle reference
                    uint32_t e = hullStart;
                    do {
                        if (hullTri[e] == (uint32_t)bl) { hullTri[e] = (uint32_t)a; break; }
                        e = hullPrev[e];
                    } while (e != hullStart);
                }
                link(a, hbl);
                link(b, halfedges[ar]);
                link(ar, bl);
                const int br = b0 + (b + 1) % 3;
                if (i < 512) stack[i++] = (uint32_t)br;
            } else {
                if (i == 0) break;
                a = (int)stack[--i];
            }
        }
        return ar;
    }
};

} // namespace

extern "C" int hg_triangulate(const float *points, int n_points, uint32_t *out_triangles, int capacity, int *n_triangles)
{
    if (!n_triangles || n_points < 0 || (n_points > 0 && !points) || capacity < 0) return HG_ERR_INVALID;
    *n_triangles = 0;
    const int n = n_points;
    std::vector<double> C(2 * (size_t)n);
    for (int i = 0; i < 2 * n; i++) { C[i] = points[i]; if (!std::isfinite(C[i])) return HG_ERR_INVALID; }
    if (n < 3) return HG_OK;
    const double INF = std::numeric_limits<double>::infinity();
    Sweep S(C);
    const size_t maxTriangles = (size_t)std::max(2 * n - 5, 0);
    S.triangles.assign(maxTriangles * 3, 0); S.halfedges.assign(maxTriangles * 3, 0);
    S.hashSize = (int)std::ceil(std::sqrt((double)n));
    S.hullPrev.assign(n, 0); S.hullNext.assign(n, 0); S.hullTri.assign(n, 0); S.hullHash.assign(S.hashSize, -1);
    std::vector<uint32_t> ids(n);
    std::vector<double> dists(n);

    double minX = INF, minY = INF, maxX = -INF, maxY = -INF;
    for (int i = 0; i < n; i++) {
        const double x = C[2 * i], y = C[2 * i + 1];
        if (x < minX) minX = x;
        if (y < minY) minY = y;
        if (x > maxX) maxX = x;
        if (y > maxY) maxY = y;
        ids[i] = (uint32_t)i;
    }
    const double cx = (minX + maxX) / 2, cy = (minY + maxY) / 2;
    double minDist = INF;
    int i0 = 0, i1 = -1, i2 = -1;
    for (int i = 0; i < n; i++) { const double d = dist2(cx, cy, C[2 * i], C[2 * i + 1]); if (d < minDist) { i0 = i; minDist = d; } }
    const double i0x = C[2 * i0], i0y = C[2 * i0 + 1];
    minDist = INF;
    for (int i = 0; i < n; i++) {
        if (i == i0) continue;
        const double d = dist2(i0x, i0y, C[2 * i], C[2 * i + 1]);
        if (d < minDist && d > 0) { i1 = i; minDist = d; }
    }
    if (i1 < 0) return HG_OK;                     // every point coincides with the seed
    double i1x = C[2 * i1], i1y = C[2 * i1 + 1];
    double minRadius = INF;
    for (int i = 0; i < n; i++) {
        if (i == i0 || i == i1) continue;
        const double r = circumradius(i0x, i0y, i1x, i1y, C[2 * i], C[2 * i + 1]);
        if (r < minRadius) { i2 = i; minRadius = r; }
    }
    if (minRadius == INF || i2 < 0) return HG_OK; // collinear input: a hull, no triangles
    double i2x = C[2 * i2], i2y = C[2 * i2 + 1];
    if (orient2d(i0x, i0y, i1x, i1y, i2x, i2y) < 0) { std::swap(i1, i2); std::swap(i1x, i2x); std::swap(i1y, i2y); }
    {
        const double dx = i1x - i0x, dy = i1y - i0y, ex = i2x - i0x, ey = i2y - i0y;
        const double bl = dx * dx + dy * dy, cl = ex * ex + ey * ey, d = 0.5 / (dx * ey - dy * ex);
        S.ccx = i0x + (ey * bl - dy * cl) * d; S.ccy = i0y + (dx * cl - ex * bl) * d;
    }
    for (int i = 0; i < n; i++) dists[i] = dist2(C[2 * i], C[2 * i + 1], S.ccx, S.ccy);
    quicksort(ids, dists, 0, n - 1);

    auto &hullPrev = S.hullPrev; auto &hullNext = S.hullNext; auto &hullTri = S.hullTri; auto &hullHash = S.hullHash;
    S.hullStart = (uint32_t)i0;
    hullNext[i0] = hullPrev[i2] = (uint32_t)i1;
    hullNext[i1] = hullPrev[i0] = (uint32_t)i2;
    hullNext[i2] = hullPrev[i1] = (uint32_t)i0;
    hullTri[i0] = 0; hullTri[i1] = 1; hullTri[i2] = 2;
    hullHash[S.hash_key(i0x, i0y)] = i0;
    hullHash[S.hash_key(i1x, i1y)] = i1;
    hullHash[S.hash_key(i2x, i2y)] = i2;
    S.add_triangle((uint32_t)i0, (uint32_t)i1, (uint32_t)i2, -1, -1, -1);

    const double EPS = 2.220446049250313e-16;     // 2^-52
    double xp = 0, yp = 0;
    for (int k = 0; k < n; k++) {
        const uint32_t i = ids[k];
        const double x = C[2 * i], y = C[2 * i + 1];
        if (k > 0 && std::fabs(x - xp) <= EPS && std::fabs(y - yp) <= EPS) continue;     // near-duplicate of the previous point
        xp = x; yp = y;
        if ((int)i == i0 || (int)i == i1 || (int)i == i2) continue;
        int start = 0;
        for (int j = 0, key = S.hash_key(x, y); j < S.hashSize; j++) {
            start = hullHash[(key + j) % S.hashSize];
            if (start != -1 && (uint32_t)start != hullNext[start]) break;
        }
        if (start == -1) start = (int)S.hullStart;
        start = (int)hullPrev[start];
        int e = start;
        uint32_t q;
        while (q = hullNext[e], orient2d(x, y, C[2 * e], C[2 * e + 1], C[2 * q], C[2 * q + 1]) >= 0) {
            e = (int)q;
            if (e == start) { e = -1; break; }
        }
        if (e == -1) continue;                    // likely a near-duplicate point
        int t = S.add_triangle((uint32_t)e, i, hullNext[e], -1, -1, (int)hullTri[e]);
        hullTri[i] = (uint32_t)S.legalize(t + 2);
        hullTri[e] = (uint32_t)t;
        uint32_t nx = hullNext[e];
        while (q = hullNext[nx], orient2d(x, y, C[2 * nx], C[2 * nx + 1], C[2 * q], C[2 * q + 1]) < 0) {
            t = S.add_triangle(nx, i, q, (int)hullTri[i], -1, (int)hullTri[nx]);
            hullTri[i] = (uint32_t)S.legalize(t + 2);
            hullNext[nx] = nx;                    // removed from the hull
            nx = q;
        }
        if (e == start) {
            while (q = hullPrev[e], orient2d(x, y, C[2 * q], C[2 * q + 1], C[2 * e], C[2 * e + 1]) < 0) {
                t = S.add_triangle(q, i, (uint32_t)e, -1, (int)hullTri[e], (int)hullTri[q]);
                S.legalize(t + 2);
                hullTri[q] = (uint32_t)t;
                hullNext[e] = (uint32_t)e;        // removed from the hull
                e = (int)q;
            }
        }
        S.hullStart = hullPrev[i] = (uint32_t)e;
        hullNext[e] = hullPrev[nx] = i;
        hullNext[i] = nx;
        hullHash[S.hash_key(x, y)] = (int32_t)i;
        hullHash[S.hash_key(C[2 * e], C[2 * e + 1])] = e;
    }
    const int total = S.trianglesLen / 3;
    *n_triangles = total;
    if (out_triangles) {
        if (total > capacity) return HG_ERR_INVALID;
        for (int k = 0; k < S.trianglesLen; k++) out_triangles[k] = S.triangles[k];
    }
    return HG_OK;
}
