// delaunay.mjs -- small host-side Delaunay triangulator (incremental Bowyer-Watson) used where the reference calls
// `new Delaunator(points).triangles` (Homography.js:1216-1218).
//
// Input: flat or nested x,y coordinates (any array / typed array).  Output: Uint32Array of vertex ids, 3 per triangle,
// the same container type delaunator@5.0.0 returns.  The reference's dependency is NOT vendored in its tree, so this is
// an independent implementation (ghost-vertex Bowyer-Watson, so the convex hull is covered exactly): it guarantees a
// valid Delaunay triangulation (empty circumcircles, cover = convex hull), not the same triangle order or the same diagonal on co-circular quads (SURVEY.md §8c: triangulation parity is
// unpinned).  O(n^2) worst case: meant for meshes of up to a few thousand landmarks, once per source-point set.

function orient(ax, ay, bx, by, cx, cy) { return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax); }

// > 0 when d is strictly inside the circumcircle of the counter-clockwise triangle a, b, c
function inCircle(ax, ay, bx, by, cx, cy, dx, dy) {
    const adx = ax - dx, ady = ay - dy, bdx = bx - dx, bdy = by - dy, cdx = cx - dx, cdy = cy - dy;
    const ad = adx * adx + ady * ady, bd = bdx * bdx + bdy * bdy, cd = cdx * cdx + cdy * cdy;
    return adx * (bdy * cd - bd * cdy) - ady * (bdx * cd - bd * cdx) + ad * (bdx * cdy - bdy * cdx);
}

// "In circumcircle" for a triangle stored counter-clockwise; the vertex id `g` is the ghost vertex at infinity: the ghost
// triangle (u, v, g) stands for the half-plane to the left of the hull edge u->v (plus the open segment u-v itself).
function conflicts(X, Y, g, a, b, c, px, py) {
    if (a !== g && b !== g && c !== g) return inCircle(X[a], Y[a], X[b], Y[b], X[c], Y[c], px, py) > 0;
    const u = a === g ? b : (b === g ? c : a), v = a === g ? c : (b === g ? a : b);
    const o = orient(X[u], Y[u], X[v], Y[v], px, py);
    if (o !== 0) return o > 0;
    return (px - X[u]) * (px - X[v]) + (py - Y[u]) * (py - Y[v]) < 0;
}

export function triangulate(points) {
    const flat = ArrayBuffer.isView(points) ? points : points.flat();
    const n = flat.length >> 1;
    if (n < 3) return new Uint32Array(0);
    const X = new Float64Array(n), Y = new Float64Array(n);
    for (let i = 0; i < n; i++) { X[i] = flat[2 * i]; Y[i] = flat[2 * i + 1]; }
    // insertion in x order (deterministic, keeps cavities small); exact duplicates are skipped
    const order = Array.from({ length: n }, (_, i) => i).sort((a, b) => (X[a] - X[b]) || (Y[a] - Y[b]) || (a - b));
    // seed: the first two distinct points and the first point not collinear with them
    const i0 = order[0];
    let k1 = 1;
    while (k1 < n && X[order[k1]] === X[i0] && Y[order[k1]] === Y[i0]) k1++;
    if (k1 >= n) return new Uint32Array(0);
    let i1 = order[k1], k2 = k1 + 1;
    while (k2 < n && orient(X[i0], Y[i0], X[i1], Y[i1], X[order[k2]], Y[order[k2]]) === 0) k2++;
    if (k2 >= n) return new Uint32Array(0);                                  // all points collinear
    let i2 = order[k2];
    if (orient(X[i0], Y[i0], X[i1], Y[i1], X[i2], Y[i2]) < 0) { const t = i1; i1 = i2; i2 = t; }
    const g = n;                                                             // ghost vertex
    let tris = [[i0, i1, i2], [i1, i0, g], [i2, i1, g], [i0, i2, g]];
    let prev = -1;
    for (const p of order) {
        const dup = prev >= 0 && X[p] === X[prev] && Y[p] === Y[prev];
        prev = p;
        if (dup || p === i0 || p === i1 || p === i2) continue;
        const px = X[p], py = Y[p];
        const keep = [], edges = new Map();
        for (const t of tris) {
            const [a, b, c] = t;
            if (conflicts(X, Y, g, a, b, c, px, py)) {
                for (const [u, v] of [[a, b], [b, c], [c, a]]) {               // interior cavity edges cancel pairwise
                    const rev = v + ',' + u;
                    if (edges.has(rev)) edges.delete(rev); else edges.set(u + ',' + v, [u, v]);
                }
            } else keep.push(t);
        }
        for (const [u, v] of edges.values()) {
            if (u === g || v === g) { keep.push([u, v, p]); continue; }      // new ghost triangle on the grown hull
            const o = orient(X[u], Y[u], X[v], Y[v], px, py);
            if (o > 0) keep.push([u, v, p]);
            else if (o < 0) keep.push([v, u, p]);
            // collinear with the cavity edge: degenerate sliver, dropped
        }
        tris = keep;
    }
    const out = [];
    for (const [a, b, c] of tris) if (a !== g && b !== g && c !== g) out.push(a, b, c);
    return Uint32Array.from(out);
}

/** Row-major split of an (nx+1) x (ny+1) point grid: (a,b,c),(b,d,c); what the benchmarks inject for regular grids. */
export function gridTriangles(nx, ny) {
    const out = new Uint32Array(nx * ny * 6), stride = nx + 1;
    let k = 0;
    for (let j = 0; j < ny; j++) for (let i = 0; i < nx; i++) {
        const a = j * stride + i;
        out[k++] = a; out[k++] = a + 1; out[k++] = a + stride; out[k++] = a + 1; out[k++] = a + stride + 1; out[k++] = a + stride;
    }
    return out;
}
