// Replays the API scripts of tests/golden/golden.json on the drop-in class (homography.js_amd/js/Homography.mjs) and
// compares with what the reference produced.
//   node tests/js/replay_golden.mjs --dry     state machine only (no GPU): path chosen, output window, points, matrices
//   node tests/js/replay_golden.mjs           full: also the warped RGBA (sha256) and the triangle map, on the GPU
// Prints one JSON line {cases, warps, failures:[...]} and exits 1 on any failure.
import fs from 'fs';
import path from 'path';
import crypto from 'crypto';
import { fileURLToPath } from 'url';
import { Homography } from '../../homography.js_amd/js/Homography.mjs';

const HERE = path.dirname(fileURLToPath(import.meta.url));
const G = JSON.parse(fs.readFileSync(path.join(HERE, '..', 'golden', 'golden.json'), 'utf8'));
const BLOBS = fs.readFileSync(path.join(HERE, '..', 'golden', 'golden_blobs.bin'));
const dry = process.argv.includes('--dry');
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
const decodePts = (p) => { if (p && p.f64blob) { const f = blobView(p.f64blob, Float64Array), o = []; for (let i = 0; i < f.length; i += 2) o.push([f[i], f[i + 1]]); return o; } return p; };
const u = (v) => (v === undefined ? null : v);

const failures = [];
let nWarps = 0, nCases = 0;
for (const c of G.cases) {
    if (only && c.name !== only.slice(7)) continue;
    if (skipBig && /_(4k|8k|1080p)/.test(c.name)) continue;
    nCases++;
    const images = {};
    for (const [k, v] of Object.entries(c.images || {})) images[k] = lcgImage(v.w, v.h, v.seed);
    const tris = c.triangles ? (c.triangles.u32blob ? blobView(c.triangles.u32blob, Uint32Array) : Uint32Array.from(c.triangles)) : new Uint32Array(0);
    Homography.triangulate = () => Uint32Array.from(tris);            // same triangles as injected into the reference
    let H = null, k = 0;
    const img = (key) => (key === null || key === undefined ? null : images[key]);
    try {
        for (const op of c.script) {
            const [name, ...a] = op;
            if (name === 'new') {
                H = new Homography(...a);
                if (dry) for (const m of ['_inverseGeometric', '_inversePiecewise', '_forwardGeometric', '_forwardPiecewise']) {
                    const pathName = { _inverseGeometric: '_inverseGeometricWarp', _inversePiecewise: '_inversePiecewiseAffineWarp', _forwardGeometric: '_geometricWarp', _forwardPiecewise: '_piecewiseAffineWarp' }[m];
                    const orig = H[m].bind(H);
                    H[m] = () => { if (m === '_inverseGeometric') H._alignRanges(); if (m === '_inversePiecewise') H._mapState = 'inverse'; H._lastPath = pathName; return new Uint8ClampedArray(0); };
                    void orig;
                }
            } else if (name === 'setSourcePoints') H.setSourcePoints(decodePts(a[0]), img(a[1]), u(a[2]), u(a[3]), u(a[4]));
            else if (name === 'setDestinyPoints') H.setDestinyPoints(decodePts(a[0]), u(a[1]));
            else if (name === 'setReferencePoints') H.setReferencePoints(decodePts(a[0]), decodePts(a[1]), img(a[2]), u(a[3]), u(a[4]), u(a[5]), u(a[6]));
            else if (name === 'setImage') H.setImage(img(a[0]), u(a[1]), u(a[2]));
            else if (name === 'setTriangles') H.setTriangles(Uint32Array.from(a[0]));
            else if (name === 'warp') {
                const w = c.warps[k++];
                nWarps++;
                const tag = `${c.name}#${k - 1}`;
                const forward = w.path === '_geometricWarp' || w.path === '_piecewiseAffineWarp';
                let out = null;
                try { out = H.warp(img(a[0]), false, !!a[1]); } catch (e) {
                    if (forward && typeof e === 'string' && e.includes('not built into this addon')) out = null; else throw e;
                }
                const chk = (cond, what) => { if (!cond) failures.push(`${tag}: ${what}`); };
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
                if (!dry && out !== null) {
                    chk(out.width === w.out.w && out.height === w.out.h, `output ${out.width}x${out.height}`);
                    chk(sha(out.data) === w.out.sha, 'RGBA sha256 differs from the reference');
                    if (w.path === '_inversePiecewiseAffineWarp') chk(sha(H.triangleMap(true)) === w.map.sha, 'triangle map differs from the reference');
                }
            }
        }
    } catch (e) {
        failures.push(`${c.name}: threw ${typeof e === 'string' ? e : (e && e.stack) || e}`);
    }
    if (H) H.close();
}
console.log(JSON.stringify({ mode: dry ? 'dry' : 'gpu', cases: nCases, warps: nWarps, failures }));
process.exit(failures.length ? 1 : 0);
