// hg_delaunay.cpp -- host Delaunay triangulator behind hg_triangulate (include/hgwarp.h).
//
// Stands where the reference calls `new Delaunator(points).triangles` (Homography.js:1216-1218 <- :262, :742).  The
// reference's dependency (delaunator 5.0.0) is not vendored and none of its tests pin the triangle order, so this is an
// independent incremental Bowyer-Watson with a ghost vertex (exact hull): valid Delaunay (empty circumcircles, cover = convex hull), output container
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
struct Tri { int a, b, c; };
struct Edge { int u, v; bool alive; };

// "In circumcircle" for a counter-clockwise triangle; vertex id g is the ghost vertex at infinity: the ghost triangle
// (u, v, g) stands for the half-plane to the left of the hull edge u->v plus the open segment u-v itself.
inline bool conflicts(const std::vector<double> &X, const std::vector<double> &Y, int g, const Tri &t, double px, double py) {
    if (t.a != g && t.b != g && t.c != g) return in_circle(X[t.a], Y[t.a], X[t.b], Y[t.b], X[t.c], Y[t.c], px, py) > 0;
    const int u = t.a == g ? t.b : (t.b == g ? t.c : t.a), v = t.a == g ? t.c : (t.b == g ? t.a : t.b);
    const double o = orient(X[u], Y[u], X[v], Y[v], px, py);
    if (o != 0) return o > 0;
    return (px - X[u]) * (px - X[v]) + (py - Y[u]) * (py - Y[v]) < 0;
}

}  // namespace

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
    std::vector<Tri> tris{{i0, i1, i2}, {i1, i0, g}, {i2, i1, g}, {i0, i2, g}}, keep;
    std::vector<Edge> edges;
    int prev = -1;
    for (int p : order) {
        const bool dup = prev >= 0 && X[p] == X[prev] && Y[p] == Y[prev];
        prev = p;
        if (dup || p == i0 || p == i1 || p == i2) continue;
        const double px = X[p], py = Y[p];
        keep.clear(); edges.clear();
        for (const Tri &t : tris) {
            if (conflicts(X, Y, g, t, px, py)) {
                const int uv[3][2] = {{t.a, t.b}, {t.b, t.c}, {t.c, t.a}};
                for (auto &e : uv) {                     // interior cavity edges cancel pairwise
                    bool cancelled = false;
                    for (Edge &o : edges) if (o.alive && o.u == e[1] && o.v == e[0]) { o.alive = false; cancelled = true; break; }
                    if (cancelled) continue;
                    bool present = false;                // re-inserting a key keeps its first position (Map.set)
                    for (Edge &o : edges) if (o.alive && o.u == e[0] && o.v == e[1]) { present = true; break; }
                    if (!present) edges.push_back({e[0], e[1], true});
                }
            } else keep.push_back(t);
        }
        for (const Edge &e : edges) {
            if (!e.alive) continue;
            if (e.u == g || e.v == g) { keep.push_back({e.u, e.v, p}); continue; }     // new ghost triangle on the grown hull
            const double o = orient(X[e.u], Y[e.u], X[e.v], Y[e.v], px, py);
            if (o > 0) keep.push_back({e.u, e.v, p});
            else if (o < 0) keep.push_back({e.v, e.u, p});
            // collinear with the cavity edge: degenerate sliver, dropped
        }
        tris.swap(keep);
    }
    int count = 0;
    for (const Tri &t : tris) if (t.a != g && t.b != g && t.c != g) {
        if (out_triangles && count < capacity) {
            out_triangles[3 * count] = (uint32_t)t.a; out_triangles[3 * count + 1] = (uint32_t)t.b; out_triangles[3 * count + 2] = (uint32_t)t.c;
        }
        count++;
    }
    *n_triangles = count;
    return (out_triangles && count > capacity) ? HG_ERR_INVALID : HG_OK;
}
