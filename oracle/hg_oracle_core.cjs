// hg_oracle_core.cjs -- JavaScript restatement of the reference's pixel loops and per-triangle solves (plain JS Numbers, single thread).
//
// TEST INFRASTRUCTURE ONLY (like oracle/hg_oracle.c): never imported by the product.  Users: oracle/hg_oracle_js.mjs (timing "the
// reference's algorithm under Node.js on this box's host cores", bench.py cpu_baseline.node) and tests/js/mock_addon.cjs (a CPU
// stand-in for the device functions of the N-API addon, so that the drop-in class's STATE MACHINE can be compared with the live
// reference call sequence by call sequence in the build container: tests/js/fuzz_ref_sequences.mjs).
// Pinned: `node oracle/hg_oracle_js.mjs check` compares every loop below with the golden vectors generated from the reference itself.
// file:line citations are into the reference's Homography.js (v1.8.0).
'use strict';

function affineFromTriangles(s, d) {                                   // :1265-1306
    const sE = s[4], sF = s[5], sA = s[0] - sE, sB = s[1] - sF, sC = s[2] - sE, sD = s[3] - sF;
    const dE = d[4], dF = d[5], dA = d[0] - dE, dB = d[1] - dF, dC = d[2] - dE, dD = d[3] - dF;
    const den = sA * sD - sB * sC;
    const iA = sD / den, iB = sB / -den, iC = sC / -den, iD = sA / den;
    const iE = (sD * sE - sC * sF) / -den, iF = (sB * sE - sA * sF) / den;
    return new Float32Array([(dA * iA) + (dC * iB), (dB * iA) + (dD * iB), (dA * iC) + (dC * iD), (dB * iC) + (dD * iD),
                             (dA * iE) + (dC * iF) + dE, (dB * iE) + (dD * iF) + dF]);
}
function inverseAffine(m) {                                            // :1345-1365
    const out = new Float32Array(6), den = m[0] * m[3] - m[1] * m[2];
    out[0] = m[3] / den; out[1] = m[1] / -den; out[2] = m[2] / -den; out[3] = m[0] / den;
    out[4] = (m[3] * m[4] - m[2] * m[5]) / -den; out[5] = (m[1] * m[4] - m[0] * m[5]) / den;
    return out;
}
function fillTriangle(t, idx, width, yOff, map) {                      // :1111-1197
    const minY = ~~Math.min(t[1], t[3], t[5]), maxY = Math.ceil(Math.max(t[1], t[3], t[5]));
    const seg = (xa, ya, xb, yb) => ({ m: xb !== xa ? (yb - ya) / (xb - xa) : Infinity, b: xb !== xa ? ya - xa * ((yb - ya) / (xb - xa)) : xa,
                                       minY: Math.min(yb, ya), maxY: Math.max(yb, ya) });
    const segs = [seg(t[0], t[1], t[2], t[3]), seg(t[0], t[1], t[4], t[5]), seg(t[2], t[3], t[4], t[5])];
    for (let y = minY; y < maxY; y++) {
        let mn = Infinity, mx = -Infinity;
        for (let i = 0; i < 3; i++) {
            const e = segs[i];
            if (y >= e.minY && y <= e.maxY) {
                let x;
                if (e.m === Infinity) x = e.b; else if (e.m === 0) continue; else x = (y - e.b) / e.m;
                if (x < mn) mn = x;
                if (x > mx) mx = x;
            }
        }
        map.fill(idx, (y - yOff) * width + Math.round(mn), (y - yOff) * width + Math.round(mx));
    }
}
const applyAffine = (m, x, y) => [(m[0] * x) + (m[2] * y) + m[4], (m[1] * x) + (m[3] * y) + m[5]];     // :1382-1385
function applyProjective(m, x, y) {                                    // :1401-1404
    const den = (m[6] * x) + (m[7] * y) + 1;
    return [((m[0] * x) + (m[1] * y) + m[2]) / den, ((m[3] * x) + (m[4] * y) + m[5]) / den];
}
// vertices of triangle i into a 6-float scratch (:792-799, :824-828, :853-856; out-of-range ids read `undefined` -> NaN)
function loadTriangle(aux, pts, tris, i) {
    for (let k = 0; k < 3; k++) { const v = tris[3 * i + k] << 1; aux[2 * k] = pts[v]; aux[2 * k + 1] = pts[v + 1]; }
    return aux;
}
/** _calculatePiecewiseAffineTransformMatrices :785-804: one forward matrix per triangle. */
function piecewiseMatrices(sp, dp, tris) {
    const out = [], aS = new Float32Array(6), aD = new Float32Array(6);
    for (let i = 0; i < tris.length / 3; i++) out.push(affineFromTriangles(loadTriangle(aS, sp, tris, i), loadTriangle(aD, dp, tris, i)));
    return out;
}
/** _buildTrianglesCorrespondencesMatrix :817-832 / _buildInverse... :845-861: Int16Array(width * height) rasterised from `pts`. */
function buildTriangleMap(pts, tris, width, height, yOff) {
    const map = new Int16Array(width * height).fill(-1), aux = new Float32Array(6);
    for (let i = 0; i < tris.length / 3; i++) fillTriangle(loadTriangle(aux, pts, tris, i), i, width, yOff, map);
    return map;
}
/** The pixel loop :1042-1056 given the map and the INVERSE matrices it indexes (a missing matrix throws, as in the reference). */
function inversePiecewiseLoop(inv, map, image, W, H, minSrcX, minSrcY, xOff, yOff, objW, objH) {
    const srcRow = W << 2, dstRow = objW << 2, out = new Uint8ClampedArray(dstRow * objH);
    for (let y = yOff; y < objH + yOff; y++) {
        for (let x = xOff; x < objW + xOff; x++) {
            const t = map[(y - yOff) * objW + (x - xOff)];
            if (t >= 0) {
                let [sx, sy] = applyAffine(inv[t], x, y);              // :1046 (a fresh 2-element array per pixel, as the reference does)
                if (sx >= minSrcX && sx < W + minSrcX && sy >= minSrcY && sy < H + minSrcY) {
                    sx = Math.round(sx); sy = Math.round(sy);
                    const si = (sy * srcRow) + (sx << 2), di = ((y - yOff) * dstRow) + ((x - xOff) << 2);
                    out[di] = image[si]; out[di + 1] = image[si + 1]; out[di + 2] = image[si + 2]; out[di + 3] = image[si + 3];
                }
            }
        }
    }
    return out;
}
/** _inversePiecewiseAffineWarp :1029-1058 from point sets (matrices and map both derived from them, the common state). */
function warpInversePiecewise(sp, dp, tris, image, W, H, minSrcX, minSrcY, xOff, yOff, objW, objH) {
    const inv = piecewiseMatrices(sp, dp, tris).map(inverseAffine);    // :785-804, :1036-1038
    const map = buildTriangleMap(dp, tris, objW, objH, yOff);          // :845-861
    return { out: inversePiecewiseLoop(inv, map, image, W, H, minSrcX, minSrcY, xOff, yOff, objW, objH), map };
}
/** _piecewiseAffineWarp :948-972 given whatever map the shared field holds (cells past its end read `undefined`) and the forward matrices. */
function forwardPiecewiseLoop(fwd, map, image, W, minSrcX, minSrcY, maxSrcX, maxSrcY, xOff, yOff, objW, objH) {
    const srcRow = W << 2, dstRow = objW << 2, mw = maxSrcX - minSrcX, out = new Uint8ClampedArray(dstRow * objH);
    for (let y = minSrcY; y < maxSrcY; y++) {
        for (let x = minSrcX; x < maxSrcX; x++) {
            const t = map[(y - minSrcY) * mw + (x - minSrcX)];
            if (t > -1) {
                const idx = (y * srcRow) + (x << 2);
                let [nx, ny] = applyAffine(fwd[t], x, y);
                nx = Math.round(nx - xOff); ny = Math.round(ny - yOff);
                const ni = (ny * dstRow) + (nx << 2);
                out[ni] = image[idx]; out[ni + 1] = image[idx + 1]; out[ni + 2] = image[idx + 2]; out[ni + 3] = image[idx + 3];
            }
        }
    }
    return out;
}
/** _inverseGeometricWarp pixel loop :997-1011; kind 0 affine (m[0..5]), 1 projective (m[0..7]); m = the INVERSE matrix of :994. */
function inverseGeometricLoop(kind, m, image, W, H, xOff, yOff, objW, objH) {
    const srcRow = W << 2, dstRow = objW << 2, out = new Uint8ClampedArray(dstRow * objH), f = kind === 0 ? applyAffine : applyProjective;
    for (let y = yOff; y < objH + yOff; y++) {
        for (let x = xOff; x < objW + xOff; x++) {
            const [sx, sy] = f(m, x, y);
            if (sx >= 0 && sx < W && sy >= 0 && sy < H) {
                const di = ((y - yOff) * dstRow) + ((x - xOff) << 2), si = (Math.round(sy) * srcRow) + (Math.round(sx) << 2);
                out[di] = image[si]; out[di + 1] = image[si + 1]; out[di + 2] = image[si + 2]; out[di + 3] = image[si + 3];
            }
        }
    }
    return out;
}
/** _geometricWarp :911-932 with the FORWARD matrix. */
function forwardGeometricLoop(kind, m, image, W, H, xOff, yOff, objW, objH) {
    const srcRow = W << 2, dstRow = objW << 2, out = new Uint8ClampedArray(dstRow * objH), f = kind === 0 ? applyAffine : applyProjective;
    for (let y = 0; y < H; y++) {
        for (let x = 0; x < W; x++) {
            const idx = (y * srcRow) + (x << 2);
            let [nx, ny] = f(m, x, y);
            nx = Math.round(nx - xOff); ny = Math.round(ny - yOff);
            const ni = (ny * dstRow) + (nx << 2);
            out[ni] = image[idx]; out[ni + 1] = image[idx + 1]; out[ni + 2] = image[idx + 2]; out[ni + 3] = image[idx + 3];
        }
    }
    return out;
}

module.exports = { affineFromTriangles, inverseAffine, fillTriangle, applyAffine, applyProjective, piecewiseMatrices, buildTriangleMap,
                   inversePiecewiseLoop, warpInversePiecewise, forwardPiecewiseLoop, inverseGeometricLoop, forwardGeometricLoop };
