// Differential sequence fuzz of the drop-in class's STATE MACHINE against the LIVE reference.  BUILD CONTAINER ONLY (needs
// /root/reference; no GPU): random API call sequences -- setters in any order, re-set source points / triangles / images, warps that
// flip between the forward and the inverse loops, warpBatch -- run op by op on the reference's own Homography.js (tests/golden/
// ref_loader.mjs) and on homography.js_amd/js/Homography.mjs over tests/js/mock_addon.cjs (the class's device calls answered by the
// JavaScript oracle's loops from exactly the arguments the class passes).  After every op: same exception or same observable state
// (transform, sizes, window, normalisation flags, both point sets in place, source bbox, null-ness of the two caches); after every
// warp: same path, same RGBA bytes.  Stale-state quirks (SURVEY.md Appendix A-Q12) are therefore compared pixel for pixel.
//   node tests/js/fuzz_ref_sequences.mjs [sequences = 300] [seed = 1] [--verbose]
// Prints one JSON line {sequences, ops, warps, stateCalls, failures: [...]}; exit 1 on any failure, 3 when the reference is absent.
import path from 'path';
import crypto from 'crypto';
import { fileURLToPath } from 'url';
import { createRequire } from 'module';
import { loadReference, referenceAvailable } from '../golden/ref_loader.mjs';
import { triangulate } from '../../homography.js_amd/js/delaunay.mjs';
import { rng, lcgImage, makeScript } from './seq_scripts.mjs';

const HERE = path.dirname(fileURLToPath(import.meta.url));
if (!referenceAvailable()) { console.log(JSON.stringify({ skipped: 'reference not present' })); process.exit(3); }
process.env.HGWARP_ADDON = path.join(HERE, 'mock_addon.cjs');
const mock = createRequire(import.meta.url)(process.env.HGWARP_ADDON);

const nSeq = Number(process.argv[2] || 300), seed0 = Number(process.argv[3] || 1), verbose = process.argv.includes('--verbose');
const sha = (t) => crypto.createHash('sha256').update(Buffer.from(t.buffer, t.byteOffset, t.byteLength)).digest('hex').slice(0, 16);

// ---------------------------------------------------------------- running a script on one side
const PATHS = { _geometricWarp: 1, _piecewiseAffineWarp: 1, _inverseGeometricWarp: 1, _inversePiecewiseAffineWarp: 1 };
function makeRunner(Cls, side) {
    let H = null, chosen = null;
    const held = [];                                         // typed arrays handed to the instance (the reference aliases and mutates them)
    const pts = (p) => { if (p && p.f32) { const a = Float32Array.from(p.f32); held.push(a); return a; } return p === null ? null : p.map((q) => q.slice()); };
    return {
        run(op, images) {
            const [name, ...a] = op;
            const img = (k) => (k === null || k === undefined ? null : images[k]);
            if (name === 'new') {
                H = new Cls(a[0], a[1], a[2]);
                if (side === 'ref') for (const w of Object.keys(PATHS)) { const orig = H[w].bind(H); H[w] = (im) => { chosen = w; return orig(im); }; }
                return null;
            }
            if (name === 'setSourcePoints') return void H.setSourcePoints(pts(a[0]), img(a[1]), a[2] === undefined ? null : a[2], a[3] === undefined ? null : a[3], a[4] === undefined ? null : a[4]);
            if (name === 'setDestinyPoints') return void H.setDestinyPoints(pts(a[0]), a[1] === undefined ? null : a[1]);
            if (name === 'setReferencePoints') return void H.setReferencePoints(pts(a[0]), pts(a[1]), img(a[2]));
            if (name === 'setImage') return void H.setImage(img(a[0]));
            if (name === 'setTriangles') return void H.setTriangles(Uint32Array.from(a[0]));
            if (name === 'warp') {
                chosen = null;
                const out = H.warp(img(a[0]), false, !!a[1]);
                return [{ path: side === 'ref' ? chosen : H._lastPath, w: out.width, h: out.height, sha: sha(out.data) }];
            }
            if (name === 'warpBatch') {                      // reference: the loop it stands for (test/benchmark.js:107-110)
                const res = [];
                if (side === 'ref') for (const d of a[0]) { chosen = null; H.setDestinyPoints(pts(d)); const out = H.warp(null, false, !!a[1]); res.push({ path: chosen, w: out.width, h: out.height, sha: sha(out.data) }); }
                else { const frames = H.warpBatch(a[0].map(pts), a[1] ? { inverse: true } : {}); frames.forEach((out) => res.push({ w: out.width, h: out.height, sha: sha(out.data) })); res[res.length - 1].path = H._lastPath; }
                return res;
            }
            throw new Error('bad op ' + name);
        },
        state() {
            if (H === null) return null;
            const f32 = (p) => (p === null || p === undefined ? null : sha(Float32Array.from(p)));
            const mapNull = side === 'ref' ? H._trianglesCorrespondencesMatrix === null : H._map === null;
            const pmNull = side === 'ref' ? H._piecewiseMatrices === null : H._pm === null;
            return { transform: H.transform, W: H._width, H: H._height, ow: H._objectiveWidth, oh: H._objectiveHeight, xo: H._xOutputOffset, yo: H._yOutputOffset,
                     sn: H._srcPointsAreNormalized, dn: H._dstPointsAreNormalized, src: f32(H._srcPoints), dst: f32(H._dstPoints),
                     bbox: [H._minSrcX, H._minSrcY, H._maxSrcX, H._maxSrcY].join(','), mapNull, pmNull,
                     tris: H._triangles === null ? null : sha(Uint32Array.from(H._triangles)), held: held.map((t) => sha(t)).join(' ') };
        },
        close() { if (H && H.close) H.close(); },
    };
}
const errRepr = (e) => (typeof e === 'string' ? 'S:' + e : (e && e.constructor ? e.constructor.name : String(e)));

(async () => {
    const ref = await loadReference();
    const { Homography: Mine } = await import('../../homography.js_amd/js/Homography.mjs');
    globalThis.__TRI__ = (p) => triangulate(p);
    Mine.triangulate = (p) => triangulate(p);
    const failures = [];
    let ops = 0, warps = 0, stateCalls = 0, throwsSeen = 0;
    // Fixed sequences in front of the random ones: corners the generator reaches too rarely.
    //  (1) a forward frame with a BLANK window (destiny x all Infinity: a NaN width passes every test of :421, so warp() takes the forward loop) over a stale
    //      inverse map that names more triangles than there are matrices: the reference's loop still walks the source bbox and throws.
    //  (2) the same with every id in range: a 1 x 1 blank frame, no exception.
    const FIXED = [];
    {
        const W = 48, Hh = 40, src = [], tri = [];
        for (let j = 0; j <= 2; j++) for (let i = 0; i <= 3; i++) src.push([i * 16, j * 20]);
        for (let j = 0; j < 2; j++) for (let i = 0; i < 3; i++) { const a = j * 4 + i; tri.push(a, a + 1, a + 4, a + 1, a + 5, a + 4); }
        const scaled = (k) => src.map(([x, y]) => [x * k, y * k]);
        const nan = src.map(([x, y]) => [Infinity, y]);              // max x = min x = Infinity: the window width is NaN, which passes every test of :421 and fails :440
        const head = [['new', 'piecewiseaffine'], ['setSourcePoints', src, 'a', W, Hh, false], ['setTriangles', tri], ['setDestinyPoints', scaled(1.25), false], ['warp']];
        FIXED.push({ images: { a: { w: W, h: Hh, seed: 77 } }, script: [...head, ['setTriangles', tri.slice(0, 3 * 8)], ['setDestinyPoints', nan, false], ['warp'], ['warp']] });
        FIXED.push({ images: { a: { w: W, h: Hh, seed: 78 } }, script: [...head, ['setDestinyPoints', nan, false], ['warp'], ['setDestinyPoints', scaled(0.95), false], ['warp']] });
        FIXED.push({ images: { a: { w: W, h: Hh, seed: 79 } }, script: [...head, ['setTriangles', tri.slice(0, 3 * 8)], ['warpBatch', [scaled(0.95), nan, scaled(0.9)], false]] });
    }
    for (let s = -FIXED.length; s < nSeq; s++) {
        const { images: specs, script } = s < 0 ? FIXED[s + FIXED.length]
                                                : makeScript(rng(seed0 * 7919 + s), { triangles: (src) => Array.from(triangulate(Float32Array.from(src.flat()))) });
        const mk = () => { const o = {}; for (const [k, v] of Object.entries(specs)) o[k] = lcgImage(v.w, v.h, v.seed); return o; };
        const A = makeRunner(ref.Homography, 'ref'), B = makeRunner(Mine, 'mine'), ia = mk(), ib = mk();
        for (let i = 0; i < script.length; i++) {
            ops++;
            let ra = null, rb = null, ea = null, eb = null;
            try { ra = A.run(script[i], ia); } catch (e) { ea = errRepr(e); }
            mock.calls.length = 0;
            try { rb = B.run(script[i], ib); } catch (e) { eb = errRepr(e); }
            stateCalls += mock.calls.filter((c) => c.endsWith('State')).length;
            const where = `seq ${s} (seed ${seed0}) op ${i} ${JSON.stringify(script[i]).slice(0, 90)}`;
            if (ea !== null || eb !== null) {
                throwsSeen++;
                // same kind of failure: identical bare strings, or the same Error class
                if (ea !== eb) { failures.push(`${where}: reference ${ea === null ? 'returned' : 'threw ' + ea}, class ${eb === null ? 'returned' : 'threw ' + eb}`); break; }
                if (ea !== null && !ea.startsWith('S:')) break;      // an Error out of the middle of a method: both instances are now in an unspecified state
            }
            const sa = JSON.stringify(A.state()), sb = JSON.stringify(B.state());
            if (sa !== sb) { failures.push(`${where}: state differs\n   ref  ${sa}\n   mine ${sb}`); break; }
            if (ra && rb) {
                warps += ra.length;
                const ja = JSON.stringify(ra), jb = JSON.stringify(rb.map((x, k) => ({ path: x.path === undefined ? ra[k].path : x.path, w: x.w, h: x.h, sha: x.sha })));
                if (ja !== jb) { failures.push(`${where}: warp differs (${mock.calls.join(',')})\n   ref  ${ja}\n   mine ${jb}`); break; }
            }
        }
        A.close(); B.close();
        if (verbose && s % 50 === 0) console.error(`  ${s} sequences, ${failures.length} failures`);
        if (failures.length >= 12) break;
    }
    ref.cleanup();
    console.log(JSON.stringify({ sequences: nSeq, ops, warps, stateCalls, throwsSeen, failures }, null, failures.length ? 1 : 0));
    process.exit(failures.length ? 1 : 0);
})().catch((e) => { console.error(e); process.exit(2); });
