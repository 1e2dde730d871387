// Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference and node >= 12.17).
//
// It imports the *reference* Homography.js (read-only, from /root/reference) through the three shims
// described in SURVEY.md Appendix B (fake `document`, `ImageData` class, Delaunator stub that returns
// harness-supplied triangles), drives it with synthetic inputs, and writes
//      tests/golden/golden.json        (inputs + expected outputs / hashes, floats as raw bit patterns)
//      tests/golden/golden_blobs.bin   (raw expected arrays for the small cases)
// Nothing of the reference's source text is written into the repository: only data (inputs/outputs).
//
//      node tests/golden/gen_golden.mjs            # regenerate everything (takes ~1-2 min)
//
// Layout of a "case": an API script (the same calls a user makes on the Homography class), the synthetic
// images it uses, the triangles injected in place of Delaunator, and for every warp() the resolved
// low-level state + outputs.  The C oracle / C-ABI tests consume the resolved state; the drop-in JS class
// tests replay the script.
import fs from 'fs';
import path from 'path';
import crypto from 'crypto';
import { fileURLToPath } from 'url';
import { loadReference } from './ref_loader.mjs';
import { rng as seqRng, makeScript } from '../js/seq_scripts.mjs';

const HERE = path.dirname(fileURLToPath(import.meta.url));
const FULL = process.env.HG_GOLDEN_FULL !== '0';       // HG_GOLDEN_FULL=0 skips the 4K/8K cases (quick run)

// ---------------------------------------------------------------- helpers
const blobs = [];
let blobOff = 0;
function blob(typed) {
    const b = Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength);
    const ref = { off: blobOff, len: b.length };
    blobs.push(Buffer.from(b)); blobOff += b.length;
    return ref;
}
// big numeric arrays go to the blob file ({off,len,dtype}); small ones stay inline
const BIG = 64;
const maybeBlob = (typed, dtype) => typed.length > BIG ? { ...blob(typed), dtype } : Array.from(typed);
const sha = (typed) => crypto.createHash('sha256').update(Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength)).digest('hex');
const f32bits = (arr) => Array.from(new Uint32Array(Float32Array.from(arr).buffer));
function f64hex(arr) {
    const d = Float64Array.from(arr), u = new Uint32Array(d.buffer), out = [];
    for (let i = 0; i < d.length; i++) out.push(u[2 * i + 1].toString(16).padStart(8, '0') + u[2 * i].toString(16).padStart(8, '0'));
    return out;
}
// Synthetic RGBA source (SURVEY.md §8d): s = s*1664525 + 1013904223 (mod 2^32); byte = s >>> 24
function lcgImage(w, h, seed) {
    const data = new Uint8ClampedArray(w * h * 4);
    let s = seed >>> 0;
    for (let i = 0; i < data.length; i++) { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; data[i] = s >>> 24; }
    return { data, width: w, height: h };
}
function solidImage(w, h) { return { data: new Uint8ClampedArray(w * h * 4).fill(255), width: w, height: h }; }
// xorshift32 for point jitter etc.
function rng(seed) { let s = (seed >>> 0) || 1; return () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s / 4294967296; }; }
// Row-major split of a (nx+1) x (ny+1) point grid: (a,b,c),(b,d,c) with a=(i,j) b=(i+1,j) c=(i,j+1) d=(i+1,j+1)
function gridTriangles(nx, ny) {
    const t = [], stride = nx + 1;
    for (let j = 0; j < ny; j++) for (let i = 0; i < nx; i++) {
        const a = j * stride + i;
        t.push(a, a + 1, a + stride, a + 1, a + stride + 1, a + stride);
    }
    return t;
}

const RAW_LIMIT = 200 * 1024;          // outputs up to this many bytes are stored raw, larger ones as sha256 only

(async () => {
const REFERENCE = await loadReference();               // the reference itself, through the three shims of SURVEY.md Appendix B
const M = REFERENCE.M;
const { Homography } = M;

// ---------------------------------------------------------------- run one API script on the reference
function runCase(c) {
    const images = {};
    for (const [k, v] of Object.entries(c.images || {})) images[k] = v.solid ? solidImage(v.w, v.h) : lcgImage(v.w, v.h, v.seed);
    globalThis.__TRI__ = () => Uint32Array.from(c.triangles || []);
    let H = null;
    const warps = [], css = [], throws = {}, opWarps = {};      // opWarps[opIndex] = [first warp record, count] of a warp / warpBatch op
    let chosen = null;
    // What the reference's two piecewise caches were computed FROM (SURVEY.md Appendix A-Q12), tracked by wrapping the methods that
    // fill them: mapDef <-> _trianglesCorrespondencesMatrix (:817-832 forward / :845-861 inverse), pmDef <-> _piecewiseMatrices (:785-804).
    let mapDef = null, pmDef = null, entryMap = null, entryPm = null;
    const f32c = (p) => Float32Array.from(p), u32c = (t) => Uint32Array.from(t);
    const samePts = (a, b) => a.length === b.length && a.every((v, i) => v === b[i] || (v !== v && b[i] !== b[i]));
    const sameTris = (a, b) => a.length === b.length && a.every((v, i) => v === b[i]);
    const img = (k) => (k === null || k === undefined) ? null : images[k];
    const pts = (p) => (p && p.f32) ? Float32Array.from(p.f32) : (p && p.undef) ? undefined : p;       // {f32:[...]} => typed array input (aliased+mutated by the reference); {undef} => undefined
    const errRepr = (e) => (typeof e === 'string' ? 'S:' + e : (e && e.constructor ? e.constructor.name : String(e)));
    function oneWarp(a0, a1) {
        chosen = null;
        const t0 = process.hrtime.bigint();
        const out = H.warp(img(a0), false, !!a1);
        const ms = Number(process.hrtime.bigint() - t0) / 1e6;
        const rec = { path: chosen, ref_ms: +ms.toFixed(3), transform: H.transform,
            W: H._width, H: H._height, objW: H._objectiveWidth, objH: H._objectiveHeight,
            xOff: H._xOutputOffset, yOff: H._yOutputOffset,
            srcNorm: H._srcPointsAreNormalized, dstNorm: H._dstPointsAreNormalized,
            srcPoints: maybeBlob(new Uint32Array(Float32Array.from(H._srcPoints).buffer), 'f32bits'), dstPoints: maybeBlob(new Uint32Array(Float32Array.from(H._dstPoints).buffer), 'f32bits'),
            out: { w: out.width, h: out.height, sha: sha(out.data) } };
        if (out.data.length <= RAW_LIMIT || c.raw) rec.out.blob = blob(out.data);
        if (H.transform === 'piecewiseaffine') {
            rec.minSrcX = H._minSrcX; rec.minSrcY = H._minSrcY; rec.maxSrcX = H._maxSrcX; rec.maxSrcY = H._maxSrcY;
            const fwd = new Float32Array(H._piecewiseMatrices.length * 6), inv = new Float32Array(fwd.length);
            H._piecewiseMatrices.forEach((m, i) => { fwd.set(m, i * 6); inv.set(M.inverseAffineMatrix(m), i * 6); });
            rec.fwdSha = sha(fwd); rec.invSha = sha(inv);
            // stale state: the loop read matrices / a map that do NOT belong to the current point sets (Q12).  Recorded explicitly:
            // the matrices as they stood (fwd blob) and, for the forward loop, the definition of the map the shared field held.
            const cur = { src: f32c(H._srcPoints), dst: f32c(H._dstPoints), tris: u32c(H._triangles) };
            const pmCurrent = entryPm !== null && samePts(entryPm.src, cur.src) && samePts(entryPm.dst, cur.dst) && sameTris(entryPm.tris, cur.tris);
            const forwardLoop = chosen === '_piecewiseAffineWarp';
            const mapUsual = !forwardLoop || (entryMap !== null && entryMap.kind === 'forward' && samePts(entryMap.pts, cur.src) && sameTris(entryMap.tris, cur.tris) &&
                                              entryMap.width === H._maxSrcX - H._minSrcX && entryMap.height === H._maxSrcY - H._minSrcY && entryMap.yOff === H._minSrcY);
            if (!pmCurrent || !mapUsual) {
                rec.stale = { matrices: !pmCurrent, map: !mapUsual, nMats: H._piecewiseMatrices.length };
                if (forwardLoop) rec.stale.mapDef = { kind: entryMap.kind, width: entryMap.width, height: entryMap.height, yOff: entryMap.yOff,
                                                      pts: maybeBlob(new Uint32Array(entryMap.pts.buffer), 'f32bits'), tris: maybeBlob(entryMap.tris, 'u32') };
            }
            if (!sameTris(cur.tris, Uint32Array.from(c.triangles || []))) rec.trisNow = maybeBlob(cur.tris, 'u32');   // (setTriangles replaced the case's list)
            if ((fwd.byteLength <= RAW_LIMIT && !c.shaOnly) || rec.stale) { rec.fwd = blob(fwd); rec.inv = blob(inv); }
            const map = H._trianglesCorrespondencesMatrix;       // int16; forward or inverse map depending on the path taken (and on what came before)
            rec.map = { len: map.length, sha: sha(map) };
            if (map.byteLength <= RAW_LIMIT || c.raw) rec.map.blob = blob(map);
        } else {
            rec.matrix = (H.transform === 'affine') ? { f32: f32bits(H._transformMatrix) } : { f64: f64hex(H._transformMatrix) };
            const invM = M.calculateTransformMatrix(H.transform, H._dstPoints, H._srcPoints);
            rec.invMatrix = (H.transform === 'affine') ? { f32: f32bits(invM) } : { f64: f64hex(invM) };
        }
        // N_hit: pixels that copied a source pixel = alpha 255 when the same call runs on an all-255 image
        if (c.nhit !== false && !rec.stale) {
            const key = a0 !== null && a0 !== undefined ? a0 : c.lastImage;
            const spec = c.images[key];
            const keep = H._image;
            H._image = solidImage(spec.w, spec.h).data;
            const o2 = H[chosen](H._image);
            let n = 0; for (let i = 3; i < o2.length; i += 4) if (o2[i] === 255) n++;
            rec.nhit = n;
            H._image = keep;
        }
        return rec;
    }
    for (let opIndex = 0; opIndex < c.script.length; opIndex++) {
        const op = c.script[opIndex];
        const [name, ...a] = op;
        const firstWarp = warps.length;
        try {
        if (name === 'new') {
            H = new Homography(...a);
            for (const w of ['_geometricWarp', '_piecewiseAffineWarp', '_inverseGeometricWarp', '_inversePiecewiseAffineWarp']) {
                const orig = H[w].bind(H);
                H[w] = (im) => { chosen = w; entryMap = mapDef; entryPm = pmDef; return orig(im); };
            }
            const after = (m, fn) => { const orig = H[m].bind(H); H[m] = (...x) => { const r = orig(...x); fn(); return r; }; };
            after('_buildTrianglesCorrespondencesMatrix', () => { mapDef = { kind: 'forward', pts: f32c(H._srcPoints), tris: u32c(H._triangles), width: H._maxSrcX - H._minSrcX, height: H._maxSrcY - H._minSrcY, yOff: H._minSrcY }; });
            after('_buildInverseTrianglesCorrespondencesMatrix', () => { mapDef = { kind: 'inverse', pts: f32c(H._dstPoints), tris: u32c(H._triangles), width: H._objectiveWidth, height: H._objectiveHeight, yOff: H._yOutputOffset }; });
            after('_calculatePiecewiseAffineTransformMatrices', () => { pmDef = { src: f32c(H._srcPoints), dst: f32c(H._dstPoints), tris: u32c(H._triangles) }; });
        } else if (name === 'setSourcePoints') H.setSourcePoints(pts(a[0]), img(a[1]), a[2] === undefined ? null : a[2], a[3] === undefined ? null : a[3], a[4] === undefined ? null : a[4]);
        else if (name === 'setDestinyPoints') H.setDestinyPoints(pts(a[0]), a[1] === undefined ? null : a[1]);
        else if (name === 'setReferencePoints') H.setReferencePoints(pts(a[0]), pts(a[1]), img(a[2]), a[3] === undefined ? null : a[3], a[4] === undefined ? null : a[4], a[5] === undefined ? null : a[5], a[6] === undefined ? null : a[6]);
        else if (name === 'setImage') H.setImage(img(a[0]), a[1] === undefined ? null : a[1], a[2] === undefined ? null : a[2]);
        else if (name === 'setTriangles') H.setTriangles(Uint32Array.from(a[0]));
        else if (name === 'css') css.push(H.getTransformationMatrixAsCSS(a[0] === undefined ? null : pts(a[0]), a[1] === undefined ? null : pts(a[1]), a[2] === undefined ? null : a[2], a[3] === undefined ? null : a[3]));
        else if (name === 'warp') warps.push(oneWarp(a[0], a[1]));
        else if (name === 'warpBatch') {                          // the caller loop the class's warpBatch() stands for (test/benchmark.js:107-110)
            for (const d of a[0]) { H.setDestinyPoints(pts(d)); warps.push({ ...oneWarp(null, a[1]), batch: opIndex }); }
        } else throw new Error('bad op ' + name);
        } catch (e) {
            if (e instanceof Error && /^bad op/.test(e.message)) throw e;
            throws[opIndex] = errRepr(e);                         // bare strings are part of the API (throw("...")); Error classes by name
            if (name === 'warp') warps.push({ throws: throws[opIndex] });
            if (typeof e !== 'string') { opWarps[opIndex] = [firstWarp, warps.length - firstWarp]; break; }     // an Error out of the middle of a method: stop the script there
        }
        if (name === 'warp' || name === 'warpBatch') opWarps[opIndex] = [firstWarp, warps.length - firstWarp];
        if (['setSourcePoints', 'setReferencePoints', 'setImage'].includes(name)) {
            const k = name === 'setSourcePoints' ? a[1] : name === 'setImage' ? a[0] : a[2];
            if (k !== null && k !== undefined) c.lastImage = k;
        }
        if (name === 'warp' && a[0] !== null && a[0] !== undefined) c.lastImage = a[0];
    }
    const { lastImage, ...rest } = c;
    // compact: big point lists in the script become {f64blob:{off,len}, n} (flat x,y doubles); big triangle lists a u32 blob
    const compactPts = (a) => ((Array.isArray(a) && a.length > BIG / 2 && Array.isArray(a[0]) && typeof a[0][0] === 'number') ? { f64blob: blob(Float64Array.from(a.flat())), n: a.length } : a);
    rest.script = c.script.map((op) => (op[0] === 'warpBatch' ? [op[0], op[1].map(compactPts), ...op.slice(2)] : op.map(compactPts)));
    if (rest.triangles && rest.triangles.length > BIG) rest.triangles = { u32blob: blob(Uint32Array.from(rest.triangles)), n: rest.triangles.length };
    const extra = {};
    if (css.length) extra.css = css;
    if (Object.keys(throws).length) extra.throws = throws;
    if (c.script.some((o) => o[0] === 'warpBatch') || Object.keys(throws).length) extra.opWarps = opWarps;
    return { ...rest, warps, ...extra };
}

const cases = [];
const add = (c) => { const t0 = Date.now(); cases.push(runCase(c)); console.error(`  ${c.name}: ${Date.now() - t0} ms`); };

// ================================================================= 1. the nodeTest flow (test/nodeTest.js:5-13) on a synthetic image
add({ name: 'nodetest_synth', images: { a: { w: 400, h: 400, seed: 1 } },
      script: [['new'], ['setReferencePoints', [[0, 0], [0, 1], [1, 0], [1, 1]], [[1 / 10, 1 / 2], [0, 1], [9 / 10, 1 / 2], [1, 1]]],
               ['setImage', 'a'], ['warp']] });

// ================================================================= 2. the 12 scenarios of test/test.js (points verbatim; image = {data,width,height})
const w = 400, h = 400;
const I = { a: { w, h, seed: 7 } };
const T9 = [0, 2, 1, 2, 3, 1, 2, 8, 3, 8, 5, 3, 1, 3, 7, 3, 4, 7, 3, 5, 4, 5, 6, 4];      // hand triangulation of test1's 3x3 points
const T4 = [0, 2, 1, 2, 3, 1];                                                           // 4 corner points (0,0),(0,h),(w,0),(w,h)
add({ name: 'test1', images: I, triangles: T9, script: [['new', 'piecewiseaffine'],
      ['setReferencePoints', [[0, 0], [0, 0.5], [0.5, 0], [0.5, 0.5], [0.5, 1], [1, 0.5], [1, 1], [0, 1], [1, 0]], [[0, 0], [0, 1], [1, 0], [1, 1], [1, 2], [2, 1], [2, 2], [0, 2], [2, 0]]],
      ['warp', 'a']] });
add({ name: 'test2', images: I, triangles: T4, script: [['new', 'piecewiseaffine'],
      ['setReferencePoints', [[0, 0], [0, 1], [1, 0], [1, 1]], [[1 / 5, 1 / 5], [0, 1 / 2], [1, 0], [1 * 6 / 8, 1 * 6 / 8]]], ['warp', 'a']] });
add({ name: 'test3', images: I, triangles: T4, script: [['new', 'piecewiseaffine'],
      ['setSourcePoints', [[0, 0], [0, h], [w, 0], [w, h]], 'a'], ['setDestinyPoints', [[0, 0], [0, 1 / 2], [1 / 2, 0], [1 / 6, 1 / 12]]], ['warp']] });
{ // test4: ctor width/height + non-square image (the DOM resize is replaced by feeding an image of the target size)
    const newW = w * 1.5, newH = h / 1.5, rW = Math.round(newW), rH = Math.round(newH);
    add({ name: 'test4', images: { a: { w: rW, h: rH, seed: 9 } }, triangles: T4, script: [['new', 'piecewiseaffine', newW, newH],
          ['setSourcePoints', [[0, 0], [0, newH], [newW, 0], [newW, newH]]], ['setImage', 'a'],
          ['setDestinyPoints', [[0, 0], [0, newH], [newW, 0], [newW * 3 / 4, newH * 3 / 4]]], ['warp']] });
}
{ // test5: sinusoidal 20x10 grid
    const sp = [], dp = [], px = 20, py = 10, amp = 20, n = 8;
    for (let y = 0; y <= h; y += h / py) for (let x = 0; x <= w; x += w / px) { sp.push([x, y]); dp.push([x, amp + y + Math.sin((x * n) / Math.PI) * amp]); }
    add({ name: 'test5', images: I, triangles: gridTriangles(px, py), script: [['new', 'piecewiseaffine', w, h], ['setImage', 'a'],
          ['setSourcePoints', sp], ['setDestinyPoints', dp], ['warp']] });
    add({ name: 'test5_inverse', images: I, triangles: gridTriangles(px, py), script: [['new', 'piecewiseaffine', w, h], ['setImage', 'a'],
          ['setSourcePoints', sp], ['setDestinyPoints', dp], ['warp', null, true]] });
}
add({ name: 'test6', images: I, script: [['new', 'affine', w, h], ['setSourcePoints', [[0, 0], [0, h], [w, 0]]],
      ['setDestinyPoints', [[100, 50], [100, h + 50], [w + 100, 50]]], ['warp', 'a']] });
add({ name: 'test6_inverse', images: I, script: [['new', 'affine', w, h], ['setSourcePoints', [[0, 0], [0, h], [w, 0]]],
      ['setDestinyPoints', [[100, 50], [100, h + 50], [w + 100, 50]]], ['warp', 'a', true]] });
add({ name: 'test7', images: I, script: [['new', 'affine'], ['setSourcePoints', [[0, 0], [0, 1], [1, 0]], null],
      ['setDestinyPoints', [[0, 1 / 2], [1 / 2, 1], [1 / 2, 0]]], ['warp', 'a']] });
add({ name: 'test7_inverse', images: I, script: [['new', 'affine'], ['setSourcePoints', [[0, 0], [0, 1], [1, 0]], null],
      ['setDestinyPoints', [[0, 1 / 2], [1 / 2, 1], [1 / 2, 0]]], ['warp', 'a', true]] });
{ // test8: 10 intermediate transforms (only the last two warps are kept as goldens to bound the file size)
    const script = [['new', 'affine'], ['setSourcePoints', [[0, 0], [0, 1], [1, 0]]]];
    for (let i = 0; i < 3; i++) { script.push(['setDestinyPoints', [[0, 0], [0, 1 / 1.25], [1 / 1.75, 0]]], ['warp', 'a'], ['setDestinyPoints', [[0, 0], [0, 1 * 1.25], [1 * 1.75, 0]]], ['warp', 'a']); }
    add({ name: 'test8', images: I, script });
}
add({ name: 'test9', images: I, script: [['new', 'affine'], ['setSourcePoints', [[0, 0], [0, 1], [1, 0]]], ['setImage', 'a'],
      ['setDestinyPoints', [[0, 0], [w, h], [w, h / 5]]], ['warp']] });
add({ name: 'test10', images: I, script: [['new', 'projective', w, h], ['setSourcePoints', [[0, 0], [0, h], [w, 0], [w, h]]],
      ['setDestinyPoints', [[0, 0], [0, h], [w, 0], [w, h]]], ['warp', 'a']] });
add({ name: 'test11', images: I, script: [['new', 'projective'], ['setSourcePoints', [[0, 0], [0, 1], [1, 0], [1, 1]], null, w, h],
      ['setDestinyPoints', [[1 - 1 / 8, 0], [1 - 1 / 8, 1], [0 + 1 / 8, 0], [0 + 1 / 8, 1]]], ['warp', 'a']] });
add({ name: 'test12', images: I, script: [['new', 'projective'], ['setSourcePoints', [[0, 0], [0, 1], [1, 1 * 2 / 10], [1, 1 * 8 / 10]]],
      ['setDestinyPoints', [[0, 1 * 2 / 10], [0, 1 * 8 / 10], [1, 0], [1, 1]]], ['warp', 'a']] });

// ================================================================= 3. BASELINE configs (SURVEY.md §8d) + small analogues with raw outputs
function cfgAffine(W, Hh) { return { src: [[0, 0], [0, Hh], [W, 0]], dst: [[0, Hh / 2], [W / 2, Hh * 8 / 10], [W / 2, 0]] }; }                  // test/benchmark.js:204-205 pattern
function cfgProjective(W, Hh) { return { src: [[0, 0], [0, Hh], [W, 0], [W, Hh]], dst: [[W / 10, 0], [W / 10, Hh], [W, Hh * 2 / 8], [W, Hh * 6 / 8]] }; }   // :282-283
function cfgSinGrid(W, Hh, nx, ny, A, n = 8) {
    const sp = [], dp = [];
    for (let j = 0; j <= ny; j++) for (let i = 0; i <= nx; i++) { const x = i * (W / nx), y = j * (Hh / ny); sp.push([x, y]); dp.push([x, A + y + Math.sin((n * x) / Math.PI) * A]); }
    return { src: sp, dst: dp, tri: gridTriangles(nx, ny) };
}
function addAffine(name, W, Hh, seed) { const c = cfgAffine(W, Hh);
    add({ name, images: { a: { w: W, h: Hh, seed } }, script: [['new', 'affine'], ['setSourcePoints', c.src, 'a', W, Hh, false], ['setDestinyPoints', c.dst, false], ['warp']] }); }
function addProjective(name, W, Hh, seed) { const c = cfgProjective(W, Hh);
    add({ name, images: { a: { w: W, h: Hh, seed } }, script: [['new', 'projective'], ['setSourcePoints', c.src, 'a', W, Hh, false], ['setDestinyPoints', c.dst, false], ['warp']] }); }
function addSin(name, W, Hh, nx, ny, A, seed, n = 8) { const c = cfgSinGrid(W, Hh, nx, ny, A, n);
    add({ name, images: { a: { w: W, h: Hh, seed } }, triangles: c.tri,
          script: [['new', 'piecewiseaffine'], ['setSourcePoints', c.src, 'a', W, Hh, false], ['setDestinyPoints', c.dst, false], ['warp']] }); }
addAffine('C1_affine_400', 400, 400, 1);
addAffine('C1_small', 80, 60, 2);
addProjective('C2_small', 160, 90, 3);
addSin('C3_small', 160, 96, 10, 10, 4, 4);
addSin('C5_small', 192, 108, 24, 12, 3, 5);
if (FULL) {
    addProjective('C2_projective_1080p', 1920, 1080, 1);
    addSin('C3_piecewise_4k', 3840, 2160, 10, 10, 40, 1);
    addSin('C3_piecewise_4k_5000tri', 3840, 2160, 50, 50, 40, 1);
    addSin('C5_piecewise_8k', 7680, 4320, 50, 50, 80, 1);
}

// ================================================================= 4. quirk cases (SURVEY.md Appendix A), all small with raw outputs
{
    // xOff > 0 and yOff > 0 (map misalignment, Q4): face-like patch in the middle of the image
    const W = 128, Hh = 96, g = cfgSinGrid(64, 48, 4, 4, 2);
    const sp = g.src.map(([x, y]) => [x + 30, y + 20]), dp = g.dst.map(([x, y]) => [x * 1.3 + 20, y * 1.4 + 9]);
    add({ name: 'quirk_xoff_pos', images: { a: { w: W, h: Hh, seed: 11 } }, triangles: g.tri,
          script: [['new', 'piecewiseaffine'], ['setSourcePoints', sp, 'a', W, Hh, false], ['setDestinyPoints', dp, false], ['warp', null, true]] });
    // xOff < 0, yOff < 0 (row-0 spans wrap to the array tail, Q3)
    const dn = g.dst.map(([x, y]) => [x * 1.5 - 7.3, y * 1.25 - 11.6]);
    add({ name: 'quirk_xoff_neg', images: { a: { w: W, h: Hh, seed: 12 } }, triangles: g.tri,
          script: [['new', 'piecewiseaffine'], ['setSourcePoints', sp, 'a', W, Hh, false], ['setDestinyPoints', dn, false], ['warp', null, true]] });
    // minSrcX < 0 : source points partly outside the image (Q10 bounds use minSrc; negative flat source indices, Q2)
    const sneg = g.src.map(([x, y]) => [x * 2.4 - 12, y * 2.3 - 9]);
    add({ name: 'quirk_minsrc_neg', images: { a: { w: W, h: Hh, seed: 13 } }, triangles: g.tri,
          script: [['new', 'piecewiseaffine'], ['setSourcePoints', sneg, 'a', W, Hh, false], ['setDestinyPoints', g.dst.map(([x, y]) => [x * 2.5, y * 2.5]), false], ['warp', null, true]] });
    // fractional .5 vertices + vertical / horizontal edges + y-offset .5 (row wrap to tail, Q3)
    const s5 = [[0, 0], [64, 0], [0, 48], [64, 48], [32, 24]], d5 = [[10.5, 0.5], [100.5, 0.5], [10.5, 80.5], [100.5, 80.5], [60.5, 30.5]];
    add({ name: 'quirk_half_vertices', images: { a: { w: W, h: Hh, seed: 14 } }, triangles: [0, 1, 4, 1, 3, 4, 3, 2, 4, 2, 0, 4],
          script: [['new', 'piecewiseaffine'], ['setSourcePoints', s5, 'a', W, Hh, false], ['setDestinyPoints', d5, false], ['warp', null, true]] });
    // folded mesh: overlapping destination triangles (last writer = max id wins) and a degenerate (zero-area) triangle
    const sf = [[0, 0], [127, 0], [0, 95], [127, 95], [64, 48], [64, 48]], df = [[0, 0], [200, 10], [20, 150], [90, 60], [150, 120], [150, 120]];
    add({ name: 'quirk_folded_degenerate', images: { a: { w: W, h: Hh, seed: 15 } }, triangles: [0, 1, 4, 1, 3, 4, 3, 2, 4, 2, 0, 4, 0, 3, 1, 4, 5, 0],
          script: [['new', 'piecewiseaffine'], ['setSourcePoints', sf, 'a', W, Hh, false], ['setDestinyPoints', df, false], ['warp', null, true]] });
    // > 32767 triangles: Int16 wrap of triangle ids (Q9): 160x110 grid => 35 200 triangles on a small image
    const gb = cfgSinGrid(160, 110, 160, 110, 1.5, 3);
    add({ name: 'quirk_int16_wrap', images: { a: { w: 160, h: 110, seed: 16 } }, triangles: gb.tri, raw: true,
          script: [['new', 'piecewiseaffine'], ['setSourcePoints', gb.src, 'a', 160, 110, false], ['setDestinyPoints', gb.dst.map(([x, y]) => [x * 1.25, y * 1.25]), false], ['warp']] });
    // projective with strong perspective (denominator crosses small values), mirrored affine, downscale
    add({ name: 'proj_strong', images: { a: { w: 120, h: 90, seed: 17 } }, script: [['new', 'projective'],
          ['setSourcePoints', [[0, 0], [0, 90], [120, 0], [120, 90]], 'a', 120, 90, false], ['setDestinyPoints', [[40, 30], [5, 170], [150, 10], [260, 200]], false], ['warp']] });
    add({ name: 'affine_mirror_down', images: { a: { w: 120, h: 90, seed: 18 } }, script: [['new', 'affine'],
          ['setSourcePoints', [[0, 0], [0, 90], [120, 0]], 'a', 120, 90, false], ['setDestinyPoints', [[70, 10], [50, 55], [10, 20]], false], ['warp']] });
    add({ name: 'affine_rot_offsets', images: { a: { w: 120, h: 90, seed: 19 } }, script: [['new', 'affine'],
          ['setSourcePoints', [[0, 0], [0, 90], [120, 0]], 'a', 120, 90, false], ['setDestinyPoints', [[-33.3, 71.7], [40.2, 150.1], [77.7, -20.4]], false], ['warp']] });
}

// ================================================================= 5. seeded fuzz: jittered grids, random scale/offset, inverse forced (raw outputs)
for (let k = 0; k < 24; k++) {
    const r = rng(1000 + k);
    const W = 24 + Math.floor(r() * 72), Hh = 16 + Math.floor(r() * 64), nx = 1 + Math.floor(r() * 6), ny = 1 + Math.floor(r() * 5);
    const sx = 0.6 + r() * 1.6, sy = 0.6 + r() * 1.6, ox = (r() - 0.3) * 40, oy = (r() - 0.3) * 30, jit = r() * 0.35;
    const sp = [], dp = [];
    for (let j = 0; j <= ny; j++) for (let i = 0; i <= nx; i++) {
        const x = i * (W / nx), y = j * (Hh / ny);
        sp.push([x, y]);
        dp.push([(x + (r() - 0.5) * jit * (W / nx)) * sx + ox, (y + (r() - 0.5) * jit * (Hh / ny)) * sy + oy]);
    }
    add({ name: `fuzz_pw_${k}`, images: { a: { w: W, h: Hh, seed: 100 + k } }, triangles: gridTriangles(nx, ny),
          script: [['new', 'piecewiseaffine'], ['setSourcePoints', sp, 'a', W, Hh, false], ['setDestinyPoints', dp, false], ['warp', null, true]] });
}
for (let k = 0; k < 12; k++) {
    const r = rng(2000 + k);
    const W = 24 + Math.floor(r() * 72), Hh = 16 + Math.floor(r() * 64);
    const q = () => [(r() - 0.2) * W * 1.6, (r() - 0.2) * Hh * 1.6];
    add({ name: `fuzz_affine_${k}`, images: { a: { w: W, h: Hh, seed: 200 + k } },
          script: [['new', 'affine'], ['setSourcePoints', [[0, 0], [0, Hh], [W, 0]], 'a', W, Hh, false], ['setDestinyPoints', [q(), q(), q()], false], ['warp', null, true]] });
    const d = [[r() * W * 0.3, r() * Hh * 0.3], [r() * W * 0.3, Hh * (0.7 + r() * 0.8)], [W * (0.7 + r() * 0.8), r() * Hh * 0.3], [W * (0.7 + r() * 0.8), Hh * (0.7 + r() * 0.8)]];
    add({ name: `fuzz_proj_${k}`, images: { a: { w: W, h: Hh, seed: 300 + k } },
          script: [['new', 'projective'], ['setSourcePoints', [[0, 0], [0, Hh], [W, 0], [W, Hh]], 'a', W, Hh, false], ['setDestinyPoints', d, false], ['warp']] });
}

// ================================================================= 5b. full-size BATCH cases: the benchmarked caller loop `for (f) { setDestinyPoints(dst_f); warp(); }`
// (test/benchmark.js:107-110) on one Homography instance.  Appended after everything else so that earlier blob offsets stay put.
// shaOnly: per-frame matrices only as SHA-256 (the blobs of 4 x 5 000 triangles would add a megabyte).
function addSinBatch(name, W, Hh, nx, ny, A, seed, ns) {
    const script = [['new', 'piecewiseaffine']];
    let tri = null;
    ns.forEach((n, k) => { const c = cfgSinGrid(W, Hh, nx, ny, A, n); tri = c.tri;
        if (k === 0) script.push(['setSourcePoints', c.src, 'a', W, Hh, false]);
        script.push(['setDestinyPoints', c.dst, false], ['warp']); });
    add({ name, images: { a: { w: W, h: Hh, seed } }, triangles: tri, shaOnly: true, script });
}
if (FULL) {
    addSinBatch('C3_batch_4k', 3840, 2160, 10, 10, 40, 1, [8, 9, 10, 11]);
    addSinBatch('C5_batch_8k', 7680, 4320, 50, 50, 80, 1, [8, 9, 10, 11]);
    // C4: 68-landmark face mesh, frames of the 512-frame orbit (inputs: tests/golden/c4_inputs.json <- gen_c4_inputs.py)
    const c4 = JSON.parse(fs.readFileSync(path.join(HERE, 'c4_inputs.json'), 'utf8'));
    const pairs = (flat) => { const o = []; for (let i = 0; i < flat.length; i += 2) o.push([flat[i], flat[i + 1]]); return o; };
    const script = [['new', 'piecewiseaffine'], ['setSourcePoints', pairs(c4.src), 'a', c4.W, c4.H, false]];
    for (const d of c4.dst) script.push(['setDestinyPoints', pairs(d), false], ['warp']);
    add({ name: 'C4_orbit_4k', images: { a: { w: c4.W, h: c4.H, seed: 1 } }, triangles: c4.triangles, shaOnly: true, orbitFrames: c4.frames, script });
}

// ================================================================= 5c. full-size FORWARD cases: what warp() dispatches to when the output is not larger
// than the source (:421 piecewise, :427 affine with an output of exactly the source size).  Appended after 5b (no blobs: shaOnly).
if (FULL) {
    // affine, half turn about the image centre: the corners map onto the same w x h box -> _geometricWarp
    for (const [name, W, Hh] of [['fwd_affine_1080p', 1920, 1080], ['fwd_affine_4k', 3840, 2160]])
        add({ name, images: { a: { w: W, h: Hh, seed: 21 } }, shaOnly: true,
              script: [['new', 'affine'], ['setSourcePoints', [[0, 0], [0, Hh], [W, 0]], 'a', W, Hh, false], ['setDestinyPoints', [[W, Hh], [W, 0], [0, Hh]], false], ['warp']] });
    // piecewise: the sinusoidal grid shrunk so that its bounding box stays inside the source size and above 1 / 1.2 of it -> _piecewiseAffineWarp
    for (const [name, W, Hh, nx, ny, A] of [['fwd_piecewise_1080p', 1920, 1080, 10, 10, 20], ['fwd_piecewise_4k', 3840, 2160, 10, 10, 40], ['fwd_piecewise_4k_dense', 3840, 2160, 32, 18, 16]]) {
        const c = cfgSinGrid(W, Hh, nx, ny, A, 8);
        const dst = c.dst.map(([x, y]) => [x * 0.95 + 7, y * 0.9 + 5]);
        add({ name, images: { a: { w: W, h: Hh, seed: 22 } }, triangles: c.tri, shaOnly: true,
              script: [['new', 'piecewiseaffine'], ['setSourcePoints', c.src, 'a', W, Hh, false], ['setDestinyPoints', dst, false], ['warp']] });
    }
}

// ================================================================= 7. call SEQUENCES over the reference's cached state (SURVEY.md Appendix A-Q12).  Appended after 5c.
// The shared map field (:819-820 / :847-848) makes a forward piecewise warp that follows an inverse one read the stale INVERSE map (:957);
// the per-triangle matrices (:769) survive setSourcePoints (once a map exists, :252-255) and setTriangles (:519) until the next
// setDestinyPoints.  Every warp below records what its loop actually read (`stale`: the matrices as they stood, the map's definition).
{
    const W = 96, Hh = 64, nx = 4, ny = 3;
    const g = cfgSinGrid(W, Hh, nx, ny, 2, 5);
    const scaled = (sx, sy, ox = 0, oy = 0, k = 5) => cfgSinGrid(W, Hh, nx, ny, 2, k).dst.map(([x, y]) => [x * sx + ox, y * sy + oy]);
    const I1 = { a: { w: W, h: Hh, seed: 31 }, b: { w: W, h: Hh, seed: 32 } };
    const head = [['new', 'piecewiseaffine'], ['setSourcePoints', g.src, 'a', W, Hh, false]];
    // inverse (output larger) -> forward (output within [1/1.2, 1] of the source): the forward loop reads the inverse map of the FIRST warp
    add({ name: 'seq_inverse_then_forward', images: I1, triangles: g.tri, script: [...head,
          ['setDestinyPoints', scaled(1.3, 1.25, 4, 3), false], ['warp'], ['setDestinyPoints', scaled(0.92, 0.9, 1, 2), false], ['warp'],
          ['setDestinyPoints', scaled(0.97, 0.88, 0, 0, 7), false], ['warp'], ['warp', 'b']] });
    // forward (fresh map) -> inverse -> forward (stale) -> inverse again -> forward (stale, the second inverse warp's map)
    add({ name: 'seq_forward_inverse_forward', images: I1, triangles: g.tri, script: [...head,
          ['setDestinyPoints', scaled(0.95, 0.93, 2, 1), false], ['warp'], ['setDestinyPoints', scaled(1.4, 1.1, -3, 5), false], ['warp'],
          ['setDestinyPoints', scaled(0.95, 0.93, 2, 1), false], ['warp'], ['setDestinyPoints', scaled(0.6, 0.7, 8, 2), false], ['warp'],
          ['setDestinyPoints', scaled(0.9, 0.99, 0, 0), false], ['warp']] });
    // the video loop (test/benchmark.js:107-110): the output bbox breathes across both thresholds of :421 in both directions
    {
        const script = [...head];
        [0.7, 0.8, 0.86, 0.95, 1.0, 1.08, 1.3, 1.05, 1.0, 0.9, 0.84, 0.8, 0.95].forEach((sc, i) => script.push(['setDestinyPoints', scaled(sc, sc * (i % 2 ? 0.97 : 1), i, 2 * i, 5 + (i % 3)), false], ['warp']));
        add({ name: 'seq_threshold_loop', images: I1, triangles: g.tri, script });
    }
    // the same kind of loop as ONE warpBatch op (what the class's warpBatch() has to equal, frame by frame)
    add({ name: 'seq_batch_mixed', images: I1, triangles: g.tri, script: [...head,
          ['warpBatch', [1.25, 0.9, 0.95, 0.7, 1.0, 0.88, 1.1, 0.93].map((sc, i) => scaled(sc, sc, i, i, 5 + (i % 2))), false],
          ['warpBatch', [0.9, 0.95, 1.0].map((sc, i) => scaled(sc, sc * 0.98, 2, 1, 6 + i)), false],
          ['warpBatch', [0.9, 1.2, 0.95].map((sc, i) => scaled(sc, sc, 0, 0, 6 + i)), true]] });
    // stale MATRICES: new source points with a map in place keep the old matrices (:252-255) -- inverse loop, then forward loop
    const src2 = g.src.map(([x, y]) => [x * 0.9 + 3, y * 0.85 + 4]);
    add({ name: 'seq_stale_matrices_source', images: I1, triangles: g.tri, script: [...head,
          ['setDestinyPoints', scaled(1.2, 1.2), false], ['warp'], ['setSourcePoints', src2], ['warp'], ['warp', null, true],
          ['setDestinyPoints', scaled(0.95, 0.95), false], ['warp'], ['setSourcePoints', g.src], ['warp']] });
    // stale matrices through setTriangles (:519): reversed order, fewer triangles; then MORE triangles than matrices: the loop throws
    const T = g.tri.length / 3, rev = [];
    for (let i = T - 1; i >= 0; i--) rev.push(g.tri[3 * i], g.tri[3 * i + 1], g.tri[3 * i + 2]);
    add({ name: 'seq_stale_matrices_triangles', images: I1, triangles: g.tri, script: [...head,
          ['setDestinyPoints', scaled(1.2, 1.15), false], ['warp'], ['setTriangles', rev], ['warp'], ['setDestinyPoints', scaled(1.1, 1.3), false], ['warp'],
          ['setTriangles', g.tri.slice(0, 3 * (T - 5))], ['warp'], ['setDestinyPoints', scaled(0.93, 0.9), false], ['warp'], ['warp', null, true]] });
    add({ name: 'seq_more_triangles_than_matrices', images: I1, triangles: g.tri.slice(0, 3 * (T - 4)), nhit: false, script: [['new', 'piecewiseaffine'],
          ['setSourcePoints', g.src, 'a', W, Hh, false], ['setDestinyPoints', scaled(1.2, 1.2), false], ['warp'], ['setTriangles', g.tri], ['warp']] });
    // a new source size drops the map (:647): the next forward warp reads a FRESH forward map again
    add({ name: 'seq_resize_resets_map', images: { ...I1, c: { w: 80, h: 72, seed: 33 } }, triangles: g.tri, script: [...head,
          ['setDestinyPoints', scaled(1.3, 1.3), false], ['warp'], ['setDestinyPoints', scaled(0.9, 0.9), false], ['warp'],
          ['setImage', 'c'], ['setDestinyPoints', scaled(0.8, 1.0), false], ['warp'], ['warp', 'a']] });
    // seeded random sequences (tests/js/seq_scripts.mjs: the generator of the live differential fuzz), piecewise with the grid's own triangles
    for (let k = 0; k < 32; k++) {
        const { images, script, grid } = makeScript(seqRng(4200 + k), { transform: 'piecewiseaffine', triangles: (src, gr) => gridTriangles(gr.nx, gr.ny) });
        add({ name: `seq_fuzz_${k}`, images, triangles: gridTriangles(grid.nx, grid.ny), nhit: false, script });
    }
    if (FULL) {                                                 // full size: C3's mesh, inverse frame then a forward frame over its stale map
        const big = (sx, sy, k) => cfgSinGrid(3840, 2160, 10, 10, 40, k).dst.map(([x, y]) => [x * sx, y * sy]);
        const c = cfgSinGrid(3840, 2160, 10, 10, 40, 8);
        add({ name: 'seq_inverse_then_forward_4k', images: { a: { w: 3840, h: 2160, seed: 1 } }, triangles: c.tri, shaOnly: true, script: [['new', 'piecewiseaffine'],
              ['setSourcePoints', c.src, 'a', 3840, 2160, false], ['setDestinyPoints', big(1, 1, 8), false], ['warp'], ['setDestinyPoints', big(0.95, 0.9, 9), false], ['warp']] });
    }
}

// ================================================================= 8. the README benchmark grid (README.md:270-328 <- test/benchmark.js:50-113, :125-190): 2 / 400-ish / ~23 000
// triangles x 200^2 / 400^2 / 800^2 outputs of a 400 x 400 source, points and call order as the benchmark's (typed-array destiny sets, first
// frame + loop frames).  Row-major grid triangles are injected in place of Delaunator.  Covers the inverse row-list regime 8192 < T <= 32767
// and the FORWARD dispatch of the 400^2 -> 400^2 rows (:421).
{
    const w = 400, h = 400;
    const nFaces = (name, pointsInX, pointsInY, outW, outH) => {
        const src = [], amplitude = 20, frame = (i) => { const d = [];
            for (let y = amplitude; y <= h - amplitude; y += h / pointsInY) for (let x = 0; x <= w; x += w / pointsInX) d.push([x * (outW / w), (y + Math.sin((x * ((i % 1) + 8)) / Math.PI) * amplitude) * (outH / h)]);
            return d; };
        let rows = 0, cols = 0;
        for (let y = amplitude; y <= h - amplitude; y += h / pointsInY) { rows++; cols = 0; for (let x = 0; x <= w; x += w / pointsInX) { cols++; src.push([x, y]); } }
        const dst0 = src.map(([x, y]) => [x * (outW / w), (y + Math.sin((x * 8) / Math.PI) * amplitude) * (outH / h)]);
        add({ name, images: { a: { w, h, seed: 41 } }, triangles: gridTriangles(cols - 1, rows - 1), shaOnly: true, nhit: false, script: [['new', 'piecewiseaffine'],
              ['setSourcePoints', src, 'a', w, h, false], ['setDestinyPoints', dst0, false], ['warp'],
              ['setDestinyPoints', { f32: frame(0).flat() }, false], ['warp'], ['setDestinyPoints', { f32: frame(1).flat() }, false], ['warp']] });
    };
    const twoFaces = (name, outW, outH) => {
        const sq = [[0, 0], [0, h], [w, 0], [w, h]], d0 = [[outW / 4, outH / 4], [0, outH / 2], [outW, 0], [outW * 5 / 8, outH * 6 / 8]];
        const mv = [[0, 0], [0, -(outH * 1 / 4) / 25], [0, 0], [(outW * 3 / 8) / 25, 0]];
        const frame = (i) => d0.map(([x, y], p) => [x + mv[p][0] * (i % 25), y + mv[p][1] * (i % 25)]).flat();
        add({ name, images: { a: { w, h, seed: 41 } }, triangles: T4, shaOnly: true, nhit: false, script: [['new', 'piecewiseaffine'],
              ['setSourcePoints', sq, 'a', w, h, false], ['setDestinyPoints', d0, false], ['warp'],
              ['setDestinyPoints', { f32: frame(0) }, false], ['warp'], ['setDestinyPoints', { f32: frame(7) }, false], ['warp'], ['setDestinyPoints', { f32: frame(24) }, false], ['warp']] });
    };
    for (const [tag, ow, oh] of [['400', w, h], ['200', w / 2, h / 2], ['800', w * 2, h * 2]]) {
        twoFaces(`readme_2tri_to_${tag}`, ow, oh);
        nFaces(`readme_20x10_to_${tag}`, 20, 10, ow, oh);
        nFaces(`readme_160x80_to_${tag}`, 160, 80, ow, oh);
    }
}

// ================================================================= 9. getTransformationMatrixAsCSS :548-587 and the API's bare-string errors, from the reference
{
    const I9 = { a: { w: 120, h: 90, seed: 51 } };
    const a3 = [[0, 0], [0, 90], [120, 0]], a3d = [[10, 5], [-20, 130], [150.5, -7.25]], n3 = [[0, 0], [0, 1], [1, 0]], n3d = [[0.1, 0.05], [0.3, 0.95], [0.8, 0.2]];
    const p4 = [[0, 0], [0, 90], [120, 0], [120, 90]], p4d = [[12, 3], [5, 101.5], [131, -4], [140, 120]], n4 = [[0, 0], [0, 1], [1, 0], [1, 1]], n4d = [[0.1, 0], [0.1, 1], [1, 0.25], [1, 0.75]];
    // affine: pixel inputs, normalised inputs, points passed to the call, width / height passed to the call, typed arrays
    add({ name: 'css_affine', images: I9, nhit: false, script: [['new', 'affine'], ['setSourcePoints', a3], ['setDestinyPoints', a3d], ['css'],
          ['css', n3, n3d], ['css', null, a3], ['css', a3, a3d, 120, 90], ['css', n3, n3d, 120, 90], ['css', { f32: a3.flat() }, { f32: a3d.flat() }],
          ['setSourcePoints', a3, 'a'], ['setDestinyPoints', a3d], ['css'], ['css', n3, null]] });
    add({ name: 'css_projective', images: I9, nhit: false, script: [['new', 'projective'], ['setSourcePoints', p4], ['setDestinyPoints', p4d], ['css'],
          ['css', n4, n4d], ['css', n4, n4d, 120, 90], ['css', p4, p4d, 120, 90], ['css', null, p4], ['css', { f32: n4.flat() }, { f32: n4d.flat() }, 400, 400],
          ['setReferencePoints', n4, n4d, 'a'], ['css'], ['css', null, p4d]] });
    add({ name: 'css_auto', images: I9, nhit: false, script: [['new'], ['css', a3, a3d], ['css', p4, p4d], ['css', n4, n4d, 64, 48], ['css', n3, n3d]] });
    // every bare string the API throws on this path (:175, :342, :413, :556-558, :584, :720, :750, :889, :1247, :1451-1474)
    const g = cfgSinGrid(120, 90, 2, 2, 2, 5), ng = g.src.map(([x, y]) => [x / 120, y / 90]);
    add({ name: 'errors_bare_strings', images: I9, triangles: g.tri, nhit: false, script: [
          ['new', 'piecewiseaffine'], ['warp'],                                                        // :413
          ['setSourcePoints', g.src], ['setDestinyPoints', g.dst.slice(0, 5)],                         // :342
          ['css'],                                                                                     // :557
          ['setDestinyPoints', g.dst], ['css'],                                                        // :558 (piecewise: no matrix)
          ['new', 'affine'], ['css'],                                                                  // :556
          ['setSourcePoints', a3.slice(0, 2)],                                                         // :1463
          ['setSourcePoints', p4],                                                                     // :1463
          ['new', 'projective'], ['setSourcePoints', a3],                                              // :1469
          ['new', 'piecewiseaffine'], ['setSourcePoints', a3.slice(0, 2)],                             // :1457
          ['new', 'auto'], ['setSourcePoints', a3.slice(0, 2)],                                        // :1451
          ['new', 'perspective'], ['setSourcePoints', a3],                                             // :1474
          ['new', 'affine'], ['setSourcePoints', n3], ['setDestinyPoints', a3d],                       // :889 (ranges cannot be aligned without a size)
          ['new', 'piecewiseaffine'], ['setSourcePoints', ng], ['setDestinyPoints', ng], ['setImage', 'a'], ['warp'],
          ['new', 'piecewiseaffine'], ['setDestinyPoints', g.dst], ['setTriangles', g.tri], ['setImage', 'a'],
          ['new', 'piecewiseaffine'], ['setSourcePoints', ng], ['setTriangles', g.tri],
          ['new', 'piecewiseaffine', 120, 90], ['setDestinyPoints', ng], ['setSourcePoints', ng], ['warp', 'a'],
          ['new'], ['setReferencePoints', { undef: true }, a3],                                              // :175
          ['new', 'projective'], ['setReferencePoints', n4, p4d], ['css'], ['setImage', 'a'], ['warp'],
          ['new', 'affine'], ['setReferencePoints', a3, n3d], ['setImage', 'a'], ['css'], ['warp'],
    ] });
}

// ================================================================= 6. per-function vectors
const func = { affine: [], inv_affine: [], projective: [], round: [], fill: [], limits: [], minmax: [] };
{
    const r = rng(42);
    const f = (s) => Math.fround((r() - 0.5) * s);
    for (let k = 0; k < 200; k++) {
        const s = k < 100 ? 2000 : 2;
        const st = [f(s), f(s), f(s), f(s), f(s), f(s)], dt = [f(s), f(s), f(s), f(s), f(s), f(s)];
        if (k % 37 === 0) { st[2] = st[0]; st[3] = st[1]; }           // degenerate source triangle => Inf/NaN entries
        const m = M.affineMatrixFromTriangles(Float32Array.from(st), Float32Array.from(dt));
        func.affine.push({ src: f32bits(st), dst: f32bits(dt), out: f32bits(m) });
        func.inv_affine.push({ m: f32bits(m), out: f32bits(M.inverseAffineMatrix(m)) });
    }
    for (let k = 0; k < 100; k++) {
        const s = k < 50 ? 2000 : 1;
        const sq = [0, 0, 0, s, s, 0, s, s].map((v) => Math.fround(v + (r() - 0.5) * s * 0.3)), dq = sq.map((v) => Math.fround(v + (r() - 0.5) * s * 0.5));
        func.projective.push({ src: f32bits(sq), dst: f32bits(dq), out: f64hex(M.projectiveMatrixFromSquares(Float32Array.from(sq), Float32Array.from(dq))) });
    }
    // pivoting-sensitive: axis-aligned rectangles (many exact zeros / ties in the 8x8 system)
    for (const [W2, H2] of [[400, 400], [1920, 1080], [1, 1]]) {
        const c = cfgProjective(W2, H2), sq = c.src.flat(), dq = c.dst.flat();
        func.projective.push({ src: f32bits(sq), dst: f32bits(dq), out: f64hex(M.projectiveMatrixFromSquares(Float32Array.from(sq), Float32Array.from(dq))) });
        func.projective.push({ src: f32bits(dq), dst: f32bits(sq), out: f64hex(M.projectiveMatrixFromSquares(Float32Array.from(dq), Float32Array.from(sq))) });
    }
    const rounds = [2.5, -2.5, -0.5, 0.5, 1.5, -1.5, 0.49999999999999994, -0.49999999999999994, 0.5000000000000001, -0.5000000000000001, 1e15 + 0.5, 4503599627370495.5,
        4503599627370496, -4503599627370495.5, 0, -0, 1e300, -1e300, Infinity, -Infinity, NaN, 2147483647.5, -2147483648.5, 3839.5, 3839.4999999999995, 123456.50000000001];
    for (let k = 0; k < 60; k++) rounds.push((r() - 0.5) * 8000, Math.floor(r() * 4000) + 0.5, -(Math.floor(r() * 4000) + 0.5));
    for (const x of rounds) func.round.push({ x: f64hex([x])[0], r: f64hex([Math.round(x)])[0] });
    // fillTriangle on small maps: vertical/horizontal edges, .5 vertices, offsets, negative starts, spans past the row end
    const tris = [[2, 1, 20, 3, 9, 14], [0, 0, 16, 0, 0, 12], [4.5, 2.5, 4.5, 10.5, 12.5, 6.5], [3, 3, 3, 3, 3, 3], [1, 2, 15, 2, 8, 2], [10, 1, 2, 9, 18, 9],
        [-6, 2, 9, 5, 2, 11], [5, -3, 12, 4, 1, 7], [30, 2, 44, 9, 25, 12], [7.25, 1.75, 13.5, 8.5, 2.125, 10.875], [0.49999, 0.5, 14.5, 0.5, 7.5, 11.4999]];
    for (let k = 0; k < 20; k++) tris.push([r() * 24 - 4, r() * 16 - 3, r() * 24 - 4, r() * 16 - 3, r() * 24 - 4, r() * 16 - 3]);
    let id = 0;
    for (const t of tris) for (const [mw, yo, rows] of [[16, 0, 12], [20, 1, 13], [16, -2, 14], [9, 3, 10]]) {
        const map = new Int16Array(mw * rows).fill(-1), t32 = Float32Array.from(t);
        M.fillTriangle(t32, (id % 7) + 1, mw, yo, map);
        func.fill.push({ tri: f32bits(t32), idx: (id % 7) + 1, width: mw, yoff: yo, rows, map: Array.from(map) });
        id++;
    }
    for (let k = 0; k < 40; k++) {
        const W2 = 10 + Math.floor(r() * 500), H2 = 10 + Math.floor(r() * 500);
        const ma = Float32Array.from([f(4), f(4), f(4), f(4), f(300), f(300)]);
        func.limits.push({ kind: 'affine', m: f32bits(ma), w: W2, h: H2, out: M.calculateTransformLimits(ma, W2, H2).map((v) => f64hex([v])[0]) });
        const mp = [1 + f(1), f(1), f(100), f(1), 1 + f(1), f(100), f(0.002), f(0.002)];
        func.limits.push({ kind: 'projective', m: f64hex(mp), w: W2, h: H2, out: M.calculateTransformLimits(mp, W2, H2).map((v) => f64hex([v])[0]) });
        const p = []; for (let i = 0; i < 10; i++) p.push(f(1000));
        func.minmax.push({ p: f32bits(p), out: M.minmaxXYofArray(Float32Array.from(p)).map((v) => f64hex([v])[0]) });
    }
}

const meta = { generator: 'tests/golden/gen_golden.mjs', reference: 'Eric-Canas/Homography.js v1.8.0 (package.json)', node: process.version,
               note: 'inputs are synthetic (LCG RGBA); triangles are injected in place of delaunator@5.0.0 (source absent => triangulation parity unpinned)',
               image_lcg: 's = s*1664525 + 1013904223 mod 2^32; byte = s >>> 24; seed per image' };
const OUT = process.env.HG_GOLDEN_OUT || HERE;             // (HG_GOLDEN_OUT: a scratch directory for trial runs)
fs.writeFileSync(path.join(OUT, 'golden.json'), JSON.stringify({ meta, cases, func }));
fs.writeFileSync(path.join(OUT, 'golden_blobs.bin'), Buffer.concat(blobs));
console.error(`wrote ${cases.length} cases, ${blobOff} blob bytes`);
REFERENCE.cleanup();
})().catch((e) => { console.error(e); process.exit(1); });
