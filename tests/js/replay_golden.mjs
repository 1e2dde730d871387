// Replays the API scripts of tests/golden/golden.json on the drop-in class (homography.js_amd/js/Homography.mjs) and
// compares with what the reference produced: path chosen, output window, points, matrices, the warped RGBA (sha256) and the
// triangle map.
//   node tests/js/replay_golden.mjs           on the GPU, through the real addon
//   node tests/js/replay_golden.mjs --dry     no GPU: the class's device calls are answered by tests/js/mock_addon.cjs (the JavaScript
//                                             oracle's loops fed with exactly the arguments the class passes) -- the class's state
//                                             machine and what it asks the native layer for, pixel for pixel
// Prints one JSON line {cases, warps, failures:[...]} and exits 1 on any failure.
import fs from 'fs';
import path from 'path';
import crypto from 'crypto';
import { fileURLToPath } from 'url';

const HERE = path.dirname(fileURLToPath(import.meta.url));
const G = JSON.parse(fs.readFileSync(path.join(HERE, '..', 'golden', 'golden.json'), 'utf8'));
const BLOBS = fs.readFileSync(path.join(HERE, '..', 'golden', 'golden_blobs.bin'));
const dry = process.argv.includes('--dry');
if (dry) {                                                // (an addon chosen by the caller -- the sanitizer build -- stays the one behind the mock's host functions)
    if (process.env.HGWARP_ADDON) process.env.HGWARP_REAL_ADDON = process.env.HGWARP_ADDON;
    process.env.HGWARP_ADDON = path.join(HERE, 'mock_addon.cjs');
}
const only = process.argv.find((a) => a.startsWith('--only='));
const skipBig = process.argv.includes('--skip-big');

const sha = (t) => crypto.createHash('sha256').update(Buffer.from(t.buffer, t.byteOffset, t.byteLength)).digest('hex');
const blobView = (ref, Ctor) => { const b = BLOBS.slice(ref.off, ref.off + ref.len); return new Ctor(b.buffer.slice(b.byteOffset, b.byteOffset + b.length)); };
function lcgImage(w, h, seed) {
    const data = new Uint8ClampedArray(w * h * 4);
    let s = seed >>> 0;
    for (let i = 0; i < data.length; i++) { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; data[i] = s >>> 24; }
    return { data, width: w, height: h };
}
const f32FromBits = (x) => new Float32Array((Array.isArray(x) ? Uint32Array.from(x) : blobView(x, Uint32Array)).buffer);
const sameF32 = (a, b) => a.length === b.length && a.every((v, i) => Object.is(v, b[i]) || v === b[i] || (Number.isNaN(v) && Number.isNaN(b[i])));
const f64FromHex = (xs) => { const d = new Float64Array(xs.length), u = new Uint32Array(d.buffer); xs.forEach((h, i) => { u[2 * i + 1] = parseInt(h.slice(0, 8), 16); u[2 * i] = parseInt(h.slice(8), 16); }); return d; };
// point arguments: {f64blob} = a big list of pairs, {f32: [...]} = a Float32Array the class aliases and mutates in place, {undef: true} = undefined
const decodePts = (p) => {
    if (p && p.f64blob) { const f = blobView(p.f64blob, Float64Array), o = []; for (let i = 0; i < f.length; i += 2) o.push([f[i], f[i + 1]]); return o; }
    if (p && p.f32) return Float32Array.from(p.f32);
    if (p && p.undef) return undefined;
    return p;
};
const u = (v) => (v === undefined ? null : v);
const errRepr = (e) => (typeof e === 'string' ? 'S:' + e : (e && e.constructor ? e.constructor.name : String(e)));

(async () => {
const { Homography } = await import('../../homography.js_amd/js/Homography.mjs');
const failures = [];
let nWarps = 0, nCases = 0, nStale = 0, nThrows = 0, nCss = 0;
for (const c of G.cases) {
    if (only && c.name !== only.slice(7)) continue;
    if (skipBig && /_(4k|8k|1080p)/.test(c.name)) continue;
    nCases++;
    const images = {};
    for (const [k, v] of Object.entries(c.images || {})) images[k] = lcgImage(v.w, v.h, v.seed);
    const tris = c.triangles ? (c.triangles.u32blob ? blobView(c.triangles.u32blob, Uint32Array) : Uint32Array.from(c.triangles)) : new Uint32Array(0);
    Homography.triangulate = () => Uint32Array.from(tris);            // same triangles as injected into the reference
    let H = null, k = 0, kc = 0;
    const img = (key) => (key === null || key === undefined ? null : images[key]);
    // one warp of the reference's record `w` against the instance's state and the returned frame (state: only when `out` is the
    // instance's latest warp -- inside a warpBatch that is the last frame)
    const checkWarp = (w, out, tag, state, tap = state) => {
        const chk = (cond, what) => { if (!cond) failures.push(`${tag}: ${what}`); };
        nWarps++;
        if (w.stale) nStale++;
        if (state) {
            chk(H._lastPath === w.path, `path ${H._lastPath} != ${w.path}`);
            chk(H.transform === w.transform, `transform ${H.transform}`);
            chk(H._width === w.W && H._height === w.H, `source size ${H._width}x${H._height} != ${w.W}x${w.H}`);
            chk(H._objectiveWidth === w.objW && H._objectiveHeight === w.objH, `objective ${H._objectiveWidth}x${H._objectiveHeight} != ${w.objW}x${w.objH}`);
            chk(H._xOutputOffset === w.xOff && H._yOutputOffset === w.yOff, `offsets ${H._xOutputOffset},${H._yOutputOffset} != ${w.xOff},${w.yOff}`);
            chk(H._srcPointsAreNormalized === w.srcNorm && H._dstPointsAreNormalized === w.dstNorm, 'normalised flags');
            chk(sameF32(Float32Array.from(H._srcPoints), f32FromBits(w.srcPoints)), 'srcPoints');
            chk(sameF32(Float32Array.from(H._dstPoints), f32FromBits(w.dstPoints)), 'dstPoints');
            if (w.transform === 'piecewiseaffine') {
                chk(H._minSrcX === w.minSrcX && H._minSrcY === w.minSrcY && H._maxSrcX === w.maxSrcX && H._maxSrcY === w.maxSrcY, 'src bbox');
            } else if (w.transform === 'affine') {
                chk(sameF32(Float32Array.from(H._transformMatrix), f32FromBits(w.matrix.f32)), 'affine matrix');
            } else {
                const want = f64FromHex(w.matrix.f64);
                chk(H._transformMatrix.length === 8 && H._transformMatrix.every((v, i) => v === want[i] || (Number.isNaN(v) && Number.isNaN(want[i]))), 'projective matrix');
            }
            // (the tap re-reads the frame the fast path prepared: not after a reference-state warp, which prepares none)
            if (tap && w.path === '_inversePiecewiseAffineWarp' && !w.stale && w.objW * w.objH >= 1) chk(sha(H.triangleMap(true)) === w.map.sha, 'triangle map differs from the reference');
        }
        chk(out.width === w.out.w && out.height === w.out.h, `output ${out.width}x${out.height} != ${w.out.w}x${w.out.h}`);
        chk(sha(out.data) === w.out.sha, `RGBA sha256 differs from the reference${w.stale ? ' (stale state: ' + JSON.stringify({ matrices: w.stale.matrices, map: w.stale.map }) + ')' : ''}`);
    };
    for (let i = 0; i < c.script.length; i++) {
        const [name, ...a] = c.script[i];
        const want = c.throws && c.throws[i] !== undefined ? c.throws[i] : null;      // what the reference threw at this op (bare string 'S:...' or an Error class)
        let got = null;
        try {
            if (name === 'new') { if (H) H.close(); H = new Homography(...a); }
            else if (name === 'setSourcePoints') H.setSourcePoints(decodePts(a[0]), img(a[1]), u(a[2]), u(a[3]), u(a[4]));
            else if (name === 'setDestinyPoints') H.setDestinyPoints(decodePts(a[0]), u(a[1]));
            else if (name === 'setReferencePoints') H.setReferencePoints(decodePts(a[0]), decodePts(a[1]), img(a[2]), u(a[3]), u(a[4]), u(a[5]), u(a[6]));
            else if (name === 'setImage') H.setImage(img(a[0]), u(a[1]), u(a[2]));
            else if (name === 'setTriangles') H.setTriangles(Uint32Array.from(a[0]));
            else if (name === 'css') {
                const str = H.getTransformationMatrixAsCSS(a[0] === undefined ? null : decodePts(a[0]), a[1] === undefined ? null : decodePts(a[1]), u(a[2]), u(a[3]));
                nCss++;
                if (str !== c.css[kc]) failures.push(`${c.name} op ${i}: CSS "${str}" != reference "${c.css[kc]}"`);
                kc++;
            } else if (name === 'warp') {
                if (c.opWarps) k = c.opWarps[i][0];
                const out = H.warp(img(a[0]), false, !!a[1]);
                checkWarp(c.warps[k++], out, `${c.name}#${k - 1}`, true);
            } else if (name === 'warpBatch') {                            // the class's batch against the reference's LOOP, frame by frame
                const [first, n] = c.opWarps[i], recs = c.warps.slice(first, first + n);
                const frames = H.warpBatch(a[0].map(decodePts), a[1] ? { inverse: true } : {});
                if (frames.length !== n) failures.push(`${c.name} op ${i}: warpBatch returned ${frames.length} frames, the reference's loop ${n}`);
                frames.forEach((out, f) => checkWarp(recs[f], out, `${c.name}#${first + f} (batch frame ${f})`, f === n - 1, false));
            }
        } catch (e) { got = errRepr(e); if (process.env.HG_REPLAY_TRACE && typeof e !== 'string') console.error(e.stack); }
        if (want !== null) nThrows++;
        if (got !== want) { failures.push(`${c.name} op ${i} ${JSON.stringify(c.script[i]).slice(0, 60)}: ${got === null ? 'returned' : 'threw ' + got}, the reference ${want === null ? 'returned' : 'threw ' + want}`); break; }
        if (got !== null && !got.startsWith('S:')) break;                 // an Error class: the reference's script stopped here too
    }
    if (H) H.close();
}
console.log(JSON.stringify({ mode: dry ? 'dry' : 'gpu', cases: nCases, warps: nWarps, staleStateWarps: nStale, expectedThrows: nThrows, cssStrings: nCss, failures }));
process.exit(failures.length ? 1 : 0);
})().catch((e) => { console.error(e); process.exit(2); });
