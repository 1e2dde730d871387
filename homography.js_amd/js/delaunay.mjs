// delaunay.mjs -- small host-side Delaunay triangulator (incremental Bowyer-Watson) used where the reference calls
// `new Delaunator(points).triangles` (Homography.js:1216-1218).
//
// Input: flat or nested x,y coordinates (any array / typed array).  Output: Uint32Array of vertex ids, 3 per triangle,
// the same container type delaunator@5.0.0 returns.  The reference's dependency is NOT vendored in its tree, so this is
// an independent implementation (ghost-vertex Bowyer-Watson, so the convex hull is covered exactly): it guarantees a
// valid Delaunay triangulation (empty circumcircles, cover = convex hull), not the same triangle order or the same diagonal on co-circular quads (SURVEY.md §8c: triangulation parity is
// unpinned).  Point location starts from the previous insertion and cavities grow through triangle adjacency: near-linear
// for the x-sorted insertion order used here.

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

// Incremental Bowyer-Watson with triangle adjacency.  Points go in in x order, so each new point lies outside (or on) the
// current hull next to the previous one: a conflicting triangle is found among the triangles created by the previous
// insertion (full scan only as a fallback), the cavity is grown through neighbours, and the new fan is stitched to the
// cavity's boundary.  V[3t + k]: vertices (counter-clockwise), N[3t + k]: the triangle across the edge opposite vertex k.
export function triangulate(points) {
    const flat = ArrayBuffer.isView(points) ? points : points.flat();
    const n = flat.length >> 1;
    if (n < 3) return new Uint32Array(0);
    const X = new Float64Array(n), Y = new Float64Array(n);
    for (let i = 0; i < n; i++) { X[i] = flat[2 * i]; Y[i] = flat[2 * i + 1]; }
    // insertion in x order (deterministic); exact duplicates are skipped
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
    const cap = 2 * n + 8;                                                   // triangles alive at any time: < 2n + 2 (ghosts included)
    const V = new Int32Array(3 * cap), N = new Int32Array(3 * cap), alive = new Uint8Array(cap), mark = new Int32Array(cap);
    const startOf = new Int32Array(n + 1), free = [];
    let count = 0, stamp = 0;
    const newTri = (a, b, c) => {
        const t = free.length ? free.pop() : count++;
        V[3 * t] = a; V[3 * t + 1] = b; V[3 * t + 2] = c; N[3 * t] = N[3 * t + 1] = N[3 * t + 2] = -1; alive[t] = 1;
        return t;
    };
    // the seed triangle and its three ghosts; N[k] is across the edge (V[k+1], V[k+2])
    const T = newTri(i0, i1, i2), G0 = newTri(i1, i0, g), G1 = newTri(i2, i1, g), G2 = newTri(i0, i2, g);
    N[3 * T] = G1; N[3 * T + 1] = G2; N[3 * T + 2] = G0;                      // across (i1,i2), (i2,i0), (i0,i1)
    N[3 * G0] = G2; N[3 * G0 + 1] = G1; N[3 * G0 + 2] = T;                    // G0 = (i1,i0,g): across (i0,g) -> G2, (g,i1) -> G1, (i1,i0) -> T
    N[3 * G1] = G0; N[3 * G1 + 1] = G2; N[3 * G1 + 2] = T;                    // G1 = (i2,i1,g): across (i1,g) -> G0, (g,i2) -> G2
    N[3 * G2] = G1; N[3 * G2 + 1] = G0; N[3 * G2 + 2] = T;                    // G2 = (i0,i2,g): across (i2,g) -> G1, (g,i0) -> G0
    let lastNew = [T, G0, G1, G2];
    const cavity = [], stack = [];
    let prev = -1;
    for (const p of order) {
        const dup = prev >= 0 && X[p] === X[prev] && Y[p] === Y[prev];
        prev = p;
        if (dup || p === i0 || p === i1 || p === i2) continue;
        const px = X[p], py = Y[p];
        const hit = (t) => conflicts(X, Y, g, V[3 * t], V[3 * t + 1], V[3 * t + 2], px, py);
        let first = -1;
        for (const t of lastNew) if (alive[t] && hit(t)) { first = t; break; }
        if (first < 0) for (let t = 0; t < count; t++) if (alive[t] && hit(t)) { first = t; break; }
        if (first < 0) continue;                                              // (numerically on top of an existing vertex)
        // cavity = connected set of conflicting triangles
        stamp++; cavity.length = 0; stack.length = 0;
        mark[first] = stamp; stack.push(first);
        while (stack.length) {
            const t = stack.pop();
            cavity.push(t);
            for (let k = 0; k < 3; k++) {
                const o = N[3 * t + k];
                if (o >= 0 && mark[o] !== stamp && hit(o)) { mark[o] = stamp; stack.push(o); }
            }
        }
        // boundary edges (a, b) of the cavity, in cavity / edge order, with the surviving triangle behind each
        const ea = [], eb = [], eo = [];
        for (const t of cavity) for (let k = 0; k < 3; k++) {
            const o = N[3 * t + k];
            if (o < 0 || mark[o] !== stamp) { ea.push(V[3 * t + (k + 1) % 3]); eb.push(V[3 * t + (k + 2) % 3]); eo.push(o); }
        }
        for (const t of cavity) { alive[t] = 0; free.push(t); }
        const fan = [];
        for (let e = 0; e < ea.length; e++) {
            const t = newTri(ea[e], eb[e], p);                                // vertex 2 = p: N[2] is across (a, b)
            fan.push(t); startOf[ea[e]] = t;
            const o = eo[e];
            N[3 * t + 2] = o;
            if (o >= 0) for (let k = 0; k < 3; k++) if (V[3 * o + (k + 1) % 3] === eb[e] && V[3 * o + (k + 2) % 3] === ea[e]) N[3 * o + k] = t;
        }
        for (let e = 0; e < fan.length; e++) {                                // stitch the fan: (a,b,p) meets (b,c,p) across (b,p)
            const t = fan[e], nx = startOf[eb[e]];
            N[3 * t] = nx;                                                    // across (b, p), opposite vertex 0 = a
            N[3 * nx + 1] = t;                                                // in (b,c,p): across (p, b), opposite vertex 1 = c
        }
        lastNew = fan;
    }
    const out = [];
    for (let t = 0; t < count; t++) if (alive[t]) {
        const a = V[3 * t], b = V[3 * t + 1], c = V[3 * t + 2];
        if (a !== g && b !== g && c !== g) out.push(a, b, c);
    }
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
