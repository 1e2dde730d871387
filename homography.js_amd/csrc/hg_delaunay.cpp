// hg_delaunay.cpp -- host Delaunay triangulator behind hg_triangulate (include/hgwarp.h).
//
// Stands where the reference calls `new Delaunator(points).triangles` (Homography.js:1216-1218 <- :262, :742).  The
// reference's dependency (delaunator 5.0.0) is not vendored and none of its tests pin the triangle order, so this is an
// independent incremental Bowyer-Watson with a ghost vertex (exact hull) and triangle adjacency: valid Delaunay (empty circumcircles, cover = convex hull), output container
// Uint32 x 3 per triangle.  It is written to give EXACTLY the same list as js/delaunay.mjs (same insertion order, same
// cavity-edge order, same double arithmetic without contraction) so the Python and JS hosts build identical meshes.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/hgwarp.h"

namespace {

inline double orient(double ax, double ay, double bx, double by, double cx, double cy) {
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
}
// > 0 when d is strictly inside the circumcircle of the counter-clockwise triangle a, b, c
inline double in_circle(double ax, double ay, double bx, double by, double cx, double cy, double dx, double dy) {
    const double adx = ax - dx, ady = ay - dy, bdx = bx - dx, bdy = by - dy, cdx = cx - dx, cdy = cy - dy;
    const double ad = adx * adx + ady * ady, bd = bdx * bdx + bdy * bdy, cd = cdx * cdx + cdy * cdy;
    return adx * (bdy * cd - bd * cdy) - ady * (bdx * cd - bd * cdx) + ad * (bdx * cdy - bdy * cdx);
}
// "In circumcircle" for a counter-clockwise triangle; vertex id g is the ghost vertex at infinity: the ghost triangle
// (u, v, g) stands for the half-plane to the left of the hull edge u->v plus the open segment u-v itself.
inline bool conflicts(const std::vector<double> &X, const std::vector<double> &Y, int g, int a, int b, int c, double px, double py) {
    if (a != g && b != g && c != g) return in_circle(X[a], Y[a], X[b], Y[b], X[c], Y[c], px, py) > 0;
    const int u = a == g ? b : (b == g ? c : a), v = a == g ? c : (b == g ? a : b);
    const double o = orient(X[u], Y[u], X[v], Y[v], px, py);
    if (o != 0) return o > 0;
    return (px - X[u]) * (px - X[v]) + (py - Y[u]) * (py - Y[v]) < 0;
}

}  // namespace

// Incremental Bowyer-Watson with triangle adjacency, step for step the algorithm of js/delaunay.mjs (same insertion order,
// same cavity / boundary / slot-reuse order), so both produce the same list.  V[3t + k]: vertices (counter-clockwise),
// N[3t + k]: the triangle across the edge opposite vertex k.
extern "C" int hg_triangulate(const float *points, int n_points, uint32_t *out_triangles, int capacity, int *n_triangles) {
    if (!n_triangles || n_points < 0 || (n_points > 0 && !points) || capacity < 0) return HG_ERR_INVALID;
    *n_triangles = 0;
    const int n = n_points;
    std::vector<double> X(n), Y(n);
    for (int i = 0; i < n; i++) {
        X[i] = points[2 * i]; Y[i] = points[2 * i + 1];
        if (!(std::isfinite(X[i]) && std::isfinite(Y[i]))) return HG_ERR_INVALID;
    }
    if (n < 3) return HG_OK;
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        if (X[a] != X[b]) return X[a] < X[b];
        if (Y[a] != Y[b]) return Y[a] < Y[b];
        return a < b;
    });
    // seed: the first two distinct points and the first point not collinear with them
    const int i0 = order[0];
    int k1 = 1;
    while (k1 < n && X[order[k1]] == X[i0] && Y[order[k1]] == Y[i0]) k1++;
    if (k1 >= n) return HG_OK;
    int i1 = order[k1], k2 = k1 + 1;
    while (k2 < n && orient(X[i0], Y[i0], X[i1], Y[i1], X[order[k2]], Y[order[k2]]) == 0) k2++;
    if (k2 >= n) return HG_OK;                           // all points collinear
    int i2 = order[k2];
    if (orient(X[i0], Y[i0], X[i1], Y[i1], X[i2], Y[i2]) < 0) std::swap(i1, i2);
    const int g = n;                                     // ghost vertex
    const size_t cap = 2 * (size_t)n + 8;
    std::vector<int> V(3 * cap), N(3 * cap), mark(cap, 0), startOf((size_t)n + 1, 0), freeList, lastNew, cavity, stack, ea, eb, eo, fan;
    std::vector<uint8_t> alive(cap, 0);
    int count = 0, stamp = 0;
    auto newTri = [&](int a, int b, int c) {
        int t;
        if (!freeList.empty()) { t = freeList.back(); freeList.pop_back(); } else t = count++;
        V[3 * t] = a; V[3 * t + 1] = b; V[3 * t + 2] = c; N[3 * t] = N[3 * t + 1] = N[3 * t + 2] = -1; alive[t] = 1;
        return t;
    };
    const int T = newTri(i0, i1, i2), G0 = newTri(i1, i0, g), G1 = newTri(i2, i1, g), G2 = newTri(i0, i2, g);
    N[3 * T] = G1; N[3 * T + 1] = G2; N[3 * T + 2] = G0;
    N[3 * G0] = G2; N[3 * G0 + 1] = G1; N[3 * G0 + 2] = T;
    N[3 * G1] = G0; N[3 * G1 + 1] = G2; N[3 * G1 + 2] = T;
    N[3 * G2] = G1; N[3 * G2 + 1] = G0; N[3 * G2 + 2] = T;
    lastNew = {T, G0, G1, G2};
    int prev = -1;
    for (int p : order) {
        const bool dup = prev >= 0 && X[p] == X[prev] && Y[p] == Y[prev];
        prev = p;
        if (dup || p == i0 || p == i1 || p == i2) continue;
        const double px = X[p], py = Y[p];
        auto hit = [&](int t) { return conflicts(X, Y, g, V[3 * t], V[3 * t + 1], V[3 * t + 2], px, py); };
        int first = -1;
        for (int t : lastNew) if (alive[t] && hit(t)) { first = t; break; }
        if (first < 0) for (int t = 0; t < count; t++) if (alive[t] && hit(t)) { first = t; break; }
        if (first < 0) continue;                         // (numerically on top of an existing vertex)
        stamp++; cavity.clear(); stack.clear();
        mark[first] = stamp; stack.push_back(first);
        while (!stack.empty()) {
            const int t = stack.back(); stack.pop_back();
            cavity.push_back(t);
            for (int k = 0; k < 3; k++) {
                const int o = N[3 * t + k];
                if (o >= 0 && mark[o] != stamp && hit(o)) { mark[o] = stamp; stack.push_back(o); }
            }
        }
        ea.clear(); eb.clear(); eo.clear();
        for (int t : cavity) for (int k = 0; k < 3; k++) {
            const int o = N[3 * t + k];
            if (o < 0 || mark[o] != stamp) { ea.push_back(V[3 * t + (k + 1) % 3]); eb.push_back(V[3 * t + (k + 2) % 3]); eo.push_back(o); }
        }
        for (int t : cavity) { alive[t] = 0; freeList.push_back(t); }
        fan.clear();
        for (size_t e = 0; e < ea.size(); e++) {
            const int t = newTri(ea[e], eb[e], p);       // vertex 2 = p: N[2] is across (a, b)
            fan.push_back(t); startOf[ea[e]] = t;
            const int o = eo[e];
            N[3 * t + 2] = o;
            if (o >= 0) for (int k = 0; k < 3; k++) if (V[3 * o + (k + 1) % 3] == eb[e] && V[3 * o + (k + 2) % 3] == ea[e]) N[3 * o + k] = t;
        }
        for (size_t e = 0; e < fan.size(); e++) {        // stitch the fan: (a,b,p) meets (b,c,p) across (b,p)
            const int t = fan[e], nx = startOf[eb[e]];
            N[3 * t] = nx;
            N[3 * nx + 1] = t;
        }
        lastNew = fan;
    }
    int total = 0;
    for (int t = 0; t < count; t++) if (alive[t]) {
        const int a = V[3 * t], b = V[3 * t + 1], c = V[3 * t + 2];
        if (a == g || b == g || c == g) continue;
        if (out_triangles && total < capacity) {
            out_triangles[3 * total] = (uint32_t)a; out_triangles[3 * total + 1] = (uint32_t)b; out_triangles[3 * total + 2] = (uint32_t)c;
        }
        total++;
    }
    *n_triangles = total;
    return (out_triangles && total > capacity) ? HG_ERR_INVALID : HG_OK;
}
