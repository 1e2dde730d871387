// delaunay.mjs -- host-side Delaunay triangulator used where the reference calls `new Delaunator(points).triangles`
// (Homography.js:1216-1218 <- :262, :742).
//
// The reference depends on delaunator@5.0.0 (+ robust-predicates@3.0.1), which is NOT vendored in its tree (package.json:11,
// CDN import Homography.js:27).  The piecewise result depends on the triangle LIST, not only on the triangulation: which
// diagonal a cocircular quad gets (every cell of a regular grid) and the order of the triangles (the largest id wins where
// spans overlap).  So this file restates delaunator 5's published algorithm step for step -- seed triangle (point nearest
// the bbox centre, its nearest neighbour, the third point of the smallest circumcircle), points sorted by distance from the
// seed circumcentre (its quicksort), advancing convex hull with the pseudo-angle hash, one new triangle per visible hull
// edge, edge-flip legalisation with its fixed stack, half-edge links -- with the same floating-point expressions
// (circumradius / circumcentre / inCircle in plain doubles, orientation through a robust predicate of which only the sign
// is used, as there).  Output: Uint32Array, 3 vertex ids per triangle, in delaunator's order.
// No test of the reference pins delaunator's output, so this remains "triangulation parity unpinned" (SURVEY.md §8c);
// hg_triangulate (csrc/hg_delaunay.cpp) is the same algorithm in C++ and returns the identical list.
//
// ATTRIBUTION.  This is a restatement (written from the published algorithm, not a copy of the source, which is absent from this
// build environment) of third-party work; see THIRD_PARTY_NOTICES.md at the repository root:
//   delaunator 5.0.0          Copyright (c) 2017, Mapbox -- ISC License                (https://github.com/mapbox/delaunator)
//   robust-predicates 3.0.1   Vladimir Agafonkin -- The Unlicense (public domain); its orient2d is a port of Jonathan R. Shewchuk's
//                             public-domain "Adaptive Precision Floating-Point Arithmetic and Fast Robust Geometric Predicates"

const EPSILON = Math.pow(2, -52);
const EDGE_STACK = new Uint32Array(512);

// ---- orientation: sign of (ay - cy) * (bx - cx) - (ax - cx) * (by - cy), exact.  Fast path + error bound as in
// robust-predicates' orient2d; inside the bound the sign comes from exact expansion arithmetic (Shewchuk's TwoDiff /
// TwoProduct / Grow-Expansion), so it is the true sign for every finite input.
const SPLITTER = 134217729;                       // 2^27 + 1
const CCW_ERR_BOUND_A = (3 + 16 * 1.1102230246251565e-16) * 1.1102230246251565e-16;
const EXP = new Float64Array(40);

function growExpansion(n, b) {                      // EXP[0..n) += b, exactly; returns the new length
    let Q = b;
    for (let i = 0; i < n; i++) {
        const e = EXP[i], x = Q + e, bv = x - Q, av = x - bv;
        EXP[i] = (Q - av) + (e - bv);
        Q = x;
    }
    EXP[n] = Q;
    return n + 1;
}
function addProduct(n, a, b, sign) {                // EXP += sign * a * b, exactly (two components)
    const x = a * b;
    let c = SPLITTER * a; const ahi = c - (c - a), alo = a - ahi;
    c = SPLITTER * b; const bhi = c - (c - b), blo = b - bhi;
    const y = alo * blo - (((x - ahi * bhi) - alo * bhi) - ahi * blo);
    n = growExpansion(n, sign * y);
    return growExpansion(n, sign * x);
}
function twoDiffTail(a, b, x) { const bv = a - x, av = x + bv; return (a - av) + (bv - b); }

export function orient2d(ax, ay, bx, by, cx, cy) {
    const detleft = (ay - cy) * (bx - cx), detright = (ax - cx) * (by - cy), det = detleft - detright;
    const detsum = Math.abs(detleft + detright);
    if (Math.abs(det) >= CCW_ERR_BOUND_A * detsum) return det;
    if (!(Math.abs(ax) < 1e150 && Math.abs(ay) < 1e150 && Math.abs(bx) < 1e150 && Math.abs(by) < 1e150 && Math.abs(cx) < 1e150 && Math.abs(cy) < 1e150)) return det;
    const p = ay - cy, q = bx - cx, r = ax - cx, s = by - cy;
    const pt = twoDiffTail(ay, cy, p), qt = twoDiffTail(bx, cx, q), rt = twoDiffTail(ax, cx, r), st = twoDiffTail(by, cy, s);
    let n = 0;
    n = addProduct(n, pt, qt, 1); n = addProduct(n, rt, st, -1);
    n = addProduct(n, p, qt, 1); n = addProduct(n, pt, q, 1); n = addProduct(n, r, st, -1); n = addProduct(n, rt, s, -1);
    n = addProduct(n, p, q, 1); n = addProduct(n, r, s, -1);
    for (let i = n - 1; i >= 0; i--) if (EXP[i] !== 0) return EXP[i] > 0 ? 1 : -1;
    return 0;
}

function pseudoAngle(dx, dy) {                      // increases monotonically with the real angle, [0..1]
    const p = dx / (Math.abs(dx) + Math.abs(dy));
    return (dy > 0 ? 3 - p : 1 + p) / 4;
}
function dist(ax, ay, bx, by) { const dx = ax - bx, dy = ay - by; return dx * dx + dy * dy; }
function inCircle(ax, ay, bx, by, cx, cy, px, py) {
    const dx = ax - px, dy = ay - py, ex = bx - px, ey = by - py, fx = cx - px, fy = cy - py;
    const ap = dx * dx + dy * dy, bp = ex * ex + ey * ey, cp = fx * fx + fy * fy;
    return dx * (ey * cp - bp * fy) - dy * (ex * cp - bp * fx) + ap * (ex * fy - ey * fx) < 0;
}
function circumradius(ax, ay, bx, by, cx, cy) {
    const dx = bx - ax, dy = by - ay, ex = cx - ax, ey = cy - ay;
    const bl = dx * dx + dy * dy, cl = ex * ex + ey * ey, d = 0.5 / (dx * ey - dy * ex);
    const x = (ey * bl - dy * cl) * d, y = (dx * cl - ex * bl) * d;
    return x * x + y * y;
}
function circumcenter(ax, ay, bx, by, cx, cy) {
    const dx = bx - ax, dy = by - ay, ex = cx - ax, ey = cy - ay;
    const bl = dx * dx + dy * dy, cl = ex * ex + ey * ey, d = 0.5 / (dx * ey - dy * ex);
    return [ax + (ey * bl - dy * cl) * d, ay + (dx * cl - ex * bl) * d];
}
function swap(arr, i, j) { const t = arr[i]; arr[i] = arr[j]; arr[j] = t; }
function quicksort(ids, dists, left, right) {
    if (right - left <= 20) {
        for (let i = left + 1; i <= right; i++) {
            const temp = ids[i], tempDist = dists[temp];
            let j = i - 1;
            while (j >= left && dists[ids[j]] > tempDist) ids[j + 1] = ids[j--];
            ids[j + 1] = temp;
        }
    } else {
        const median = (left + right) >> 1;
        let i = left + 1, j = right;
        swap(ids, median, i);
        if (dists[ids[left]] > dists[ids[right]]) swap(ids, left, right);
        if (dists[ids[i]] > dists[ids[right]]) swap(ids, i, right);
        if (dists[ids[left]] > dists[ids[i]]) swap(ids, left, i);
        const temp = ids[i], tempDist = dists[temp];
        while (true) {
            do i++; while (dists[ids[i]] < tempDist);
            do j--; while (dists[ids[j]] > tempDist);
            if (j < i) break;
            swap(ids, i, j);
        }
        ids[left + 1] = ids[j];
        ids[j] = temp;
        if (right - i + 1 >= j - left) { quicksort(ids, dists, i, right); quicksort(ids, dists, left, j - 1); }
        else { quicksort(ids, dists, left, j - 1); quicksort(ids, dists, i, right); }
    }
}

export function triangulate(points) {
    const coords = ArrayBuffer.isView(points) ? points : Float64Array.from(points.flat());
    const n = coords.length >> 1;
    for (let i = 0; i < 2 * n; i++) if (!Number.isFinite(coords[i])) throw ('hgwarp: triangulate() needs finite coordinates');
    if (n < 3) return new Uint32Array(0);
    const maxTriangles = Math.max(2 * n - 5, 0);
    const triangles = new Uint32Array(maxTriangles * 3), halfedges = new Int32Array(maxTriangles * 3);
    const hashSize = Math.ceil(Math.sqrt(n));
    const hullPrev = new Uint32Array(n), hullNext = new Uint32Array(n), hullTri = new Uint32Array(n), hullHash = new Int32Array(hashSize).fill(-1);
    const ids = new Uint32Array(n), dists = new Float64Array(n);
    let trianglesLen = 0, hullStart = 0, ccx = 0, ccy = 0;

    const link = (a, b) => { halfedges[a] = b; if (b !== -1) halfedges[b] = a; };
    const addTriangle = (i0, i1, i2, a, b, c) => {
        const t = trianglesLen;
        triangles[t] = i0; triangles[t + 1] = i1; triangles[t + 2] = i2;
        link(t, a); link(t + 1, b); link(t + 2, c);
        trianglesLen += 3;
        return t;
    };
    // a point exactly on the circumcentre has no angle (NaN in delaunator, which then reads outside its hash): bucket 0 here
    const hashKey = (x, y) => { const k = Math.floor(pseudoAngle(x - ccx, y - ccy) * hashSize) % hashSize; return k === k ? k : 0; };
    const legalize = (a) => {
        let i = 0, ar = 0;
        while (true) {                              // recursion eliminated with a fixed-size stack
            const b = halfedges[a];
            // if the pair of triangles sharing edge a|b violates the Delaunay condition (p1 inside the circumcircle of
            // [p0, pl, pr]) flip the edge, then check the two edges on the far side of the flipped pair
            const a0 = a - a % 3;
            ar = a0 + (a + 2) % 3;
            if (b === -1) {                         // convex hull edge
                if (i === 0) break;
                a = EDGE_STACK[--i];
                continue;
            }
            const b0 = b - b % 3, al = a0 + (a + 1) % 3, bl = b0 + (b + 2) % 3;
            const p0 = triangles[ar], pr = triangles[a], pl = triangles[al], p1 = triangles[bl];
            const illegal = inCircle(coords[2 * p0], coords[2 * p0 + 1], coords[2 * pr], coords[2 * pr + 1],
                                     coords[2 * pl], coords[2 * pl + 1], coords[2 * p1], coords[2 * p1 + 1]);
            if (illegal) {
                triangles[a] = p1;
                triangles[b] = p0;
                const hbl = halfedges[bl];
                if (hbl === -1) {                   // edge swapped on the other side of the hull (rare): fix the hull's triangle reference
                    let e = hullStart;
                    do {
                        if (hullTri[e] === bl) { hullTri[e] = a; break; }
                        e = hullPrev[e];
                    } while (e !== hullStart);
                }
                link(a, hbl);
                link(b, halfedges[ar]);
                link(ar, bl);
                const br = b0 + (b + 1) % 3;
                if (i < EDGE_STACK.length) EDGE_STACK[i++] = br;
            } else {
                if (i === 0) break;
                a = EDGE_STACK[--i];
            }
        }
        return ar;
    };

    // bbox centre, seed point, its nearest neighbour, third point of the smallest circumcircle
    let minX = Infinity, minY = Infinity, maxX = -Infinity, maxY = -Infinity;
    for (let i = 0; i < n; i++) {
        const x = coords[2 * i], y = coords[2 * i + 1];
        if (x < minX) minX = x;
        if (y < minY) minY = y;
        if (x > maxX) maxX = x;
        if (y > maxY) maxY = y;
        ids[i] = i;
    }
    const cx = (minX + maxX) / 2, cy = (minY + maxY) / 2;
    let minDist = Infinity, i0 = 0, i1 = -1, i2 = -1;
    for (let i = 0; i < n; i++) {
        const d = dist(cx, cy, coords[2 * i], coords[2 * i + 1]);
        if (d < minDist) { i0 = i; minDist = d; }
    }
    const i0x = coords[2 * i0], i0y = coords[2 * i0 + 1];
    minDist = Infinity;
    for (let i = 0; i < n; i++) {
        if (i === i0) continue;
        const d = dist(i0x, i0y, coords[2 * i], coords[2 * i + 1]);
        if (d < minDist && d > 0) { i1 = i; minDist = d; }
    }
    if (i1 < 0) return new Uint32Array(0);          // every point coincides with the seed
    let i1x = coords[2 * i1], i1y = coords[2 * i1 + 1];
    let minRadius = Infinity;
    for (let i = 0; i < n; i++) {
        if (i === i0 || i === i1) continue;
        const r = circumradius(i0x, i0y, i1x, i1y, coords[2 * i], coords[2 * i + 1]);
        if (r < minRadius) { i2 = i; minRadius = r; }
    }
    if (minRadius === Infinity || i2 < 0) return new Uint32Array(0);        // collinear input: a hull, no triangles
    let i2x = coords[2 * i2], i2y = coords[2 * i2 + 1];
    if (orient2d(i0x, i0y, i1x, i1y, i2x, i2y) < 0) {                      // seed triangle in the orientation the hull walk expects
        const i = i1, x = i1x, y = i1y;
        i1 = i2; i1x = i2x; i1y = i2y;
        i2 = i; i2x = x; i2y = y;
    }
    [ccx, ccy] = circumcenter(i0x, i0y, i1x, i1y, i2x, i2y);
    for (let i = 0; i < n; i++) dists[i] = dist(coords[2 * i], coords[2 * i + 1], ccx, ccy);
    quicksort(ids, dists, 0, n - 1);

    hullStart = i0;
    hullNext[i0] = hullPrev[i2] = i1;
    hullNext[i1] = hullPrev[i0] = i2;
    hullNext[i2] = hullPrev[i1] = i0;
    hullTri[i0] = 0; hullTri[i1] = 1; hullTri[i2] = 2;
    hullHash[hashKey(i0x, i0y)] = i0;
    hullHash[hashKey(i1x, i1y)] = i1;
    hullHash[hashKey(i2x, i2y)] = i2;
    addTriangle(i0, i1, i2, -1, -1, -1);

    for (let k = 0, xp = 0, yp = 0; k < n; k++) {
        const i = ids[k], x = coords[2 * i], y = coords[2 * i + 1];
        if (k > 0 && Math.abs(x - xp) <= EPSILON && Math.abs(y - yp) <= EPSILON) continue;      // near-duplicate of the previous point
        xp = x; yp = y;
        if (i === i0 || i === i1 || i === i2) continue;
        // a visible hull edge through the angular hash
        let start = 0;
        for (let j = 0, key = hashKey(x, y); j < hashSize; j++) {
            start = hullHash[(key + j) % hashSize];
            if (start !== -1 && start !== hullNext[start]) break;
        }
        if (start === -1) start = hullStart;       // (delaunator would index with -1 here; cannot happen while the hull has entries)
        start = hullPrev[start];
        let e = start, q;
        while (q = hullNext[e], orient2d(x, y, coords[2 * e], coords[2 * e + 1], coords[2 * q], coords[2 * q + 1]) >= 0) {
            e = q;
            if (e === start) { e = -1; break; }
        }
        if (e === -1) continue;                     // likely a near-duplicate point
        // first triangle from the point, then flip until the Delaunay condition holds
        let t = addTriangle(e, i, hullNext[e], -1, -1, hullTri[e]);
        hullTri[i] = legalize(t + 2);
        hullTri[e] = t;
        // walk forward through the hull, adding triangles and flipping
        let nx = hullNext[e];
        while (q = hullNext[nx], orient2d(x, y, coords[2 * nx], coords[2 * nx + 1], coords[2 * q], coords[2 * q + 1]) < 0) {
            t = addTriangle(nx, i, q, hullTri[i], -1, hullTri[nx]);
            hullTri[i] = legalize(t + 2);
            hullNext[nx] = nx;                      // removed from the hull
            nx = q;
        }
        // walk backward from the other side
        if (e === start) {
            while (q = hullPrev[e], orient2d(x, y, coords[2 * q], coords[2 * q + 1], coords[2 * e], coords[2 * e + 1]) < 0) {
                t = addTriangle(q, i, e, -1, hullTri[e], hullTri[q]);
                legalize(t + 2);
                hullTri[q] = t;
                hullNext[e] = e;                    // removed from the hull
                e = q;
            }
        }
        hullStart = hullPrev[i] = e;
        hullNext[e] = hullPrev[nx] = i;
        hullNext[i] = nx;
        hullHash[hashKey(x, y)] = i;
        hullHash[hashKey(coords[2 * e], coords[2 * e + 1])] = e;
    }
    return triangles.slice(0, trianglesLen);
}

/** Row-major split of a regular (nx+1) x (ny+1) point grid: (a,b,c),(b,d,c) with a=(i,j) b=(i+1,j) c=(i,j+1) d=(i+1,j+1). */
export function gridTriangles(nx, ny) {
    const t = new Uint32Array(nx * ny * 6), stride = nx + 1;
    let k = 0;
    for (let j = 0; j < ny; j++) for (let i = 0; i < nx; i++) {
        const a = j * stride + i;
        t[k++] = a; t[k++] = a + 1; t[k++] = a + stride; t[k++] = a + 1; t[k++] = a + stride + 1; t[k++] = a + stride;
    }
    return t;
}
