// delaunay.mjs -- small host-side Delaunay triangulator (incremental Bowyer-Watson) used where the reference calls
// `new Delaunator(points).triangles` (Homography.js:1216-1218).
//
// Input: flat or nested x,y coordinates (any array / typed array).  Output: Uint32Array of vertex ids, 3 per triangle,
// the same container type delaunator@5.0.0 returns.  The reference's dependency is NOT vendored in its tree, so this is
// an independent implementation: it guarantees a valid Delaunay triangulation (empty circumcircles, cover = convex
// hull), not the same triangle order or the same diagonal on co-circular quads (SURVEY.md §8c: triangulation parity is
// unpinned).  O(n^2) worst case: meant for meshes of up to a few thousand landmarks, once per source-point set.

function orient(ax, ay, bx, by, cx, cy) { return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax); }

// > 0 when d is strictly inside the circumcircle of the counter-clockwise triangle a, b, c
function inCircle(ax, ay, bx, by, cx, cy, dx, dy) {
    const adx = ax - dx, ady = ay - dy, bdx = bx - dx, bdy = by - dy, cdx = cx - dx, cdy = cy - dy;
    const ad = adx * adx + ady * ady, bd = bdx * bdx + bdy * bdy, cd = cdx * cdx + cdy * cdy;
    return adx * (bdy * cd - bd * cdy) - ady * (bdx * cd - bd * cdx) + ad * (bdx * cdy - bdy * cdx);
}

export function triangulate(points) {
    const flat = ArrayBuffer.isView(points) ? points : points.flat();
    const n = flat.length >> 1;
    if (n < 3) return new Uint32Array(0);
    let minX = Infinity, minY = Infinity, maxX = -Infinity, maxY = -Infinity;
    for (let i = 0; i < n; i++) {
        const x = flat[2 * i], y = flat[2 * i + 1];
        if (x < minX) minX = x; if (x > maxX) maxX = x; if (y < minY) minY = y; if (y > maxY) maxY = y;
    }
    const span = Math.max(maxX - minX, maxY - minY, 1e-9), cx = (minX + maxX) / 2, cy = (minY + maxY) / 2;
    // vertex coordinates incl. a super-triangle (ids n, n+1, n+2) far outside the data
    const X = new Float64Array(n + 3), Y = new Float64Array(n + 3);
    for (let i = 0; i < n; i++) { X[i] = flat[2 * i]; Y[i] = flat[2 * i + 1]; }
    const R = 64 * span;
    X[n] = cx - R; Y[n] = cy - R; X[n + 1] = cx + R; Y[n + 1] = cy - R; X[n + 2] = cx; Y[n + 2] = cy + R;
    let tris = [[n, n + 1, n + 2]];                      // counter-clockwise
    // insert in x order (keeps the cavity search cheap enough and deterministic)
    const order = Array.from({ length: n }, (_, i) => i).sort((a, b) => (X[a] - X[b]) || (Y[a] - Y[b]) || (a - b));
    let prev = -1;
    for (const p of order) {
        if (prev >= 0 && X[p] === X[prev] && Y[p] === Y[prev]) continue;    // exact duplicates are skipped
        prev = p;
        const px = X[p], py = Y[p];
        const keep = [], edges = new Map();
        for (const t of tris) {
            const [a, b, c] = t;
            if (inCircle(X[a], Y[a], X[b], Y[b], X[c], Y[c], px, py) > 0) {
                for (const [u, v] of [[a, b], [b, c], [c, a]]) {
                    const rev = v + ',' + u;
                    if (edges.has(rev)) edges.delete(rev); else edges.set(u + ',' + v, [u, v]);
                }
            } else keep.push(t);
        }
        for (const [u, v] of edges.values()) {
            if (orient(X[u], Y[u], X[v], Y[v], px, py) > 0) keep.push([u, v, p]);
            else if (orient(X[u], Y[u], X[v], Y[v], px, py) < 0) keep.push([v, u, p]);
            // collinear with the cavity edge: degenerate sliver, dropped
        }
        tris = keep;
    }
    const out = [];
    for (const [a, b, c] of tris) if (a < n && b < n && c < n) out.push(a, b, c);
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
