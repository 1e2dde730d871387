// mock_addon.cjs -- a CPU stand-in for the DEVICE functions of lib/hgwarp.node, for ONE purpose: comparing the drop-in class's state
// machine (homography.js_amd/js/Homography.mjs) with the live reference call sequence by call sequence in the build container, where
// there is no GPU (tests/js/fuzz_ref_sequences.mjs; select it with HGWARP_ADDON=<this file>).
// TEST INFRASTRUCTURE ONLY.  The host-side functions (solves, limits, min/max, triangulation) are the REAL addon's -- they need no GPU
// --; every function that would launch a kernel is replaced by the JavaScript oracle's loop (oracle/hg_oracle_core.cjs, pinned against
// the reference's goldens) fed with exactly the arguments the class hands to the native side.  What this proves: the class asks the
// native layer for the right thing.  That the native layer then computes it bit-exactly is what the `-m gpu` parity tests prove.
'use strict';
const path = require('path');
const real = require(process.env.HGWARP_REAL_ADDON || path.join(__dirname, '..', '..', 'homography.js_amd', 'lib', 'hgwarp.node'));
const core = require(path.join(__dirname, '..', '..', 'oracle', 'hg_oracle_core.cjs'));

const calls = [];                                            // names of the device entry points hit, in order (the fuzzer reads and clears it)
const log = (n) => { calls.push(n); };
const imageOf = (c, f) => (c.images ? c.images[f % c.images.length] : c.image);
const needImage = (c) => { if (!c.image && !c.images) throw ('hgwarp mock: no source image'); };
const matsOf = (flat) => { const out = []; for (let i = 0; i < flat.length; i += 6) out.push(flat.subarray(i, i + 6)); return out; };
function checkIds(map, n, cells) {                           // a map id without a matrix: the reference throws a TypeError at that pixel
    for (let i = 0; i < cells; i++) if (map[i] > -1 && map[i] >= n) throw new TypeError("Cannot read property '0' of undefined");
}

const mock = {
    calls,
    create: () => ({ image: null, images: null, W: 0, H: 0, mesh: null, frame: null }),
    destroy: () => {},
    deviceCount: () => 1,
    setImage(c, data, w, h) { c.image = Uint8ClampedArray.from(data); c.images = null; c.W = w; c.H = h; },
    setImages(c, datas, w, h) { c.images = datas.map((d) => Uint8ClampedArray.from(d)); c.image = null; c.W = w; c.H = h; },
    piecewiseSetMesh(c, src, tris, minX, minY) { c.mesh = { src: Float32Array.from(src), tris: Uint32Array.from(tris), minX, minY }; },
    piecewisePrepare(c, dst, xo, yo, ow, oh) { c.frame = { dst: Float32Array.from(dst), xo, yo, ow, oh }; },
    warpInversePiecewise(c) {
        log('warpInversePiecewise'); needImage(c);
        const m = c.mesh, f = c.frame;
        const r = core.warpInversePiecewise(m.src, f.dst, m.tris, imageOf(c, 0), c.W, c.H, m.minX, m.minY, f.xo, f.yo, f.ow, f.oh);
        c.lastMap = r.map;
        return r.out;
    },
    getTriMap: (c) => c.lastMap,
    getMatrices(c, T) {
        const fwd = core.piecewiseMatrices(c.mesh.src, c.frame.dst, c.mesh.tris), F = new Float32Array(6 * T), I = new Float32Array(6 * T);
        fwd.forEach((mm, i) => { F.set(mm, 6 * i); I.set(core.inverseAffine(mm), 6 * i); });
        return { forward: F, inverse: I };
    },
    warpInversePiecewiseBatch(c, pts, g, own, images, w, h) {
        log('warpInversePiecewiseBatch'); if (images) mock.setImages(c, images, w, h); needImage(c);
        const m = c.mesh, n = m.src.length, out = [];
        for (let k = 0; k < g.length / 4; k++)
            out.push(core.warpInversePiecewise(m.src, pts.subarray(k * n, (k + 1) * n), m.tris, imageOf(c, k), c.W, c.H, m.minX, m.minY, g[4 * k], g[4 * k + 1], g[4 * k + 2], g[4 * k + 3]).out);
        return out;
    },
    warpForwardPiecewise(c, dst, maxX, maxY, xo, yo, ow, oh) {
        log('warpForwardPiecewise'); needImage(c);
        return mock._forward(c, dst, maxX, maxY, xo, yo, ow, oh, 0);
    },
    _forward(c, dst, maxX, maxY, xo, yo, ow, oh, f) {
        const m = c.mesh, map = core.buildTriangleMap(m.src, m.tris, maxX - m.minX, maxY - m.minY, m.minY);
        return core.forwardPiecewiseLoop(core.piecewiseMatrices(m.src, dst, m.tris), map, imageOf(c, f), c.W, m.minX, m.minY, maxX, maxY, xo, yo, ow, oh);
    },
    warpForwardPiecewiseBatch(c, pts, maxX, maxY, g, own, images, w, h) {
        log('warpForwardPiecewiseBatch'); if (images) mock.setImages(c, images, w, h); needImage(c);
        const n = c.mesh.src.length, out = [];
        for (let k = 0; k < g.length / 4; k++) out.push(mock._forward(c, pts.subarray(k * n, (k + 1) * n), maxX, maxY, g[4 * k], g[4 * k + 1], g[4 * k + 2], g[4 * k + 3], k));
        return out;
    },
    // reference-state forms (stale matrices / stale map, SURVEY.md Appendix A-Q12)
    warpInversePiecewiseState(c, mats, dst, tris, minX, minY, xo, yo, ow, oh) {
        log('warpInversePiecewiseState'); needImage(c);
        const fwd = matsOf(mats), map = core.buildTriangleMap(dst, tris, ow, oh, yo);
        checkIds(map, fwd.length, map.length);
        return core.inversePiecewiseLoop(fwd.map(core.inverseAffine), map, imageOf(c, 0), c.W, c.H, minX, minY, xo, yo, ow, oh);
    },
    warpForwardPiecewiseState(c, mats, mapPts, mapTris, mw, mh, myOff, minX, minY, maxX, maxY, xo, yo, ow, oh) {
        log('warpForwardPiecewiseState'); needImage(c);
        const fwd = matsOf(mats), map = mw * mh >= 1 ? core.buildTriangleMap(mapPts, mapTris, mw, mh, myOff) : new Int16Array(0);
        const cells = (maxX - minX) * (maxY - minY);
        checkIds(map, fwd.length, Math.min(map.length, cells > 0 ? cells : 0));
        return core.forwardPiecewiseLoop(fwd, map, imageOf(c, 0), c.W, minX, minY, maxX, maxY, xo, yo, ow, oh);
    },
    warpInverseGeometric(c, kind, inv, xo, yo, ow, oh) {
        log('warpInverseGeometric'); needImage(c);
        return core.inverseGeometricLoop(kind, inv, imageOf(c, 0), c.W, c.H, xo, yo, ow, oh);
    },
    warpInverseGeometricBatch(c, kind, from, to, g, own, images, w, h) {
        log('warpInverseGeometricBatch'); if (images) mock.setImages(c, images, w, h); needImage(c);
        const per = kind === 0 ? 6 : 8, out = [];
        for (let k = 0; k < g.length / 4; k++) {
            const a = from.subarray(k * per, (k + 1) * per), b = to.subarray(k * per, (k + 1) * per);
            const m = kind === 0 ? real.solveAffine(a, b) : real.solveProjective(a, b);
            out.push(core.inverseGeometricLoop(kind, m, imageOf(c, k), c.W, c.H, g[4 * k], g[4 * k + 1], g[4 * k + 2], g[4 * k + 3]));
        }
        return out;
    },
    warpForwardGeometric(c, kind, m, xo, yo, ow, oh) {
        log('warpForwardGeometric'); needImage(c);
        return core.forwardGeometricLoop(kind, m, imageOf(c, 0), c.W, c.H, xo, yo, ow, oh);
    },
    warpForwardGeometricBatch(c, kind, mats, g, own, images, w, h) {
        log('warpForwardGeometricBatch'); if (images) mock.setImages(c, images, w, h); needImage(c);
        const out = [];
        for (let k = 0; k < g.length / 4; k++) out.push(core.forwardGeometricLoop(kind, mats.subarray(8 * k, 8 * k + 8), imageOf(c, k), c.W, c.H, g[4 * k], g[4 * k + 1], g[4 * k + 2], g[4 * k + 3]));
        return out;
    },
    // several devices, one host thread: the frames of a device list are the frames of one device
    multiCreate: () => mock.create(), multiDestroy: () => {},
    multiSetImage(m, data, w, h) { mock.setImage(m, data, w, h); },
    multiSetMesh(m, src, tris, minX, minY) { mock.piecewiseSetMesh(m, src, tris, minX, minY); },
    multiWarpBatch(m, pts, g, datas, w, h) { return mock.warpInversePiecewiseBatch(m, pts, g, false, datas, w, h); },
    multiWarpGeometricBatch(m, kind, from, to, g, datas, w, h) { return mock.warpInverseGeometricBatch(m, kind, from, to, g, false, datas, w, h); },
    // frame pool: plain V8 arrays here
    pinnedBuffer: (n) => new Uint8ClampedArray(n),
    poolPressure: () => false, poolCollected: () => {}, release: () => {}, releaseBatch: () => {}, setPinnedLimit: () => 0, poolStats: () => ({}),
};
for (const k of ['solveAffine', 'invertAffine', 'solveProjective', 'transformLimits', 'minmaxXY', 'triangulate', 'solveAffineTriangles'])
    if (real[k]) mock[k] = real[k];
if (!mock.solveAffineTriangles) throw new Error('hgwarp.node lacks solveAffineTriangles: rebuild it (make -C homography.js_amd)');
module.exports = mock;
