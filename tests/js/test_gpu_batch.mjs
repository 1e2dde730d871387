// GPU: warpBatch() (the setDestinyPoints+warp loop as one pass) equals per-frame warp(…, applyAlwaysInverse = true),
// repeated warps on one instance, mutation of the caller's image buffer is seen by the next warp (the reference aliases it).
import crypto from 'crypto';
import { Homography } from '../../homography.js_amd/js/Homography.mjs';
import { gridTriangles } from '../../homography.js_amd/js/delaunay.mjs';

const fails = [];
const ok = (c, m) => { if (!c) fails.push(m); };
const sha = (t) => crypto.createHash('sha256').update(Buffer.from(t.buffer, t.byteOffset, t.byteLength)).digest('hex');
function lcgImage(w, h, seed) {
    const data = new Uint8ClampedArray(w * h * 4);
    let s = seed >>> 0;
    for (let i = 0; i < data.length; i++) { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; data[i] = s >>> 24; }
    return { data, width: w, height: h };
}
const W = 512, H = 288, nx = 8, ny = 6, A = 9;
const src = [], sets = [];
for (let j = 0; j <= ny; j++) for (let i = 0; i <= nx; i++) src.push([i * W / nx, j * H / ny]);
for (let f = 0; f < 5; f++) sets.push(src.map(([x, y]) => [x * (1 + 0.1 * f), A + y + Math.sin(((8 + f) * x) / Math.PI) * A]));
const img = lcgImage(W, H, 21);
Homography.triangulate = () => gridTriangles(nx, ny);
const h = new Homography('piecewiseaffine');
h.setSourcePoints(src, img, W, H, false);
const single = sets.map((d) => { h.setDestinyPoints(d, false); return h.warp(null, false, true); });
const batch = h.warpBatch(sets);
ok(batch.length === sets.length, 'batch length');
batch.forEach((b, f) => {
    ok(b.width === single[f].width && b.height === single[f].height, `frame ${f} size`);
    ok(sha(b.data) === sha(single[f].data), `frame ${f} differs between warpBatch and warp`);
    ok(b.data instanceof Uint8ClampedArray && b.data.length === 4 * b.width * b.height, `frame ${f} type`);
});
// the caller mutates its buffer in place: the next warp must see it (reference re-reads image.data every warp, :298)
h.setDestinyPoints(sets[0], false);
const before = sha(h.warp(null, false, true).data);
img.data.fill(7);
const after = h.warp(null, false, true);
ok(sha(after.data) !== before, 'mutated source buffer was not picked up');
ok(after.data.every((v) => v === 7 || v === 0), 'mutated source: output should only hold 7 or 0');
// affine + projective on the same instance type, output identical across repeated calls
const g = new Homography('projective');
g.setReferencePoints([[0, 0], [0, 1], [1, 0], [1, 1]], [[1 / 10, 1 / 2], [0, 1], [9 / 10, 1 / 2], [1, 1]]);
const img2 = lcgImage(400, 400, 1);
const o1 = g.warp(img2), o2 = g.warp();
ok(o1.width === 400 && o1.height === 200 && sha(o1.data) === sha(o2.data), 'projective repeat');
{   // opt-in reuseOutput: same bytes, one buffer behind every returned frame, sized views
    const r = new Homography('piecewiseaffine', null, null, { reuseOutput: true });
    r.setSourcePoints(src, lcgImage(W, H, 21), W, H, false);
    const seen = new Set();
    for (let pass = 0; pass < 2; pass++) sets.forEach((d, f) => {       // (the frames grow with f: pass 0 sizes the buffer)
        r.setDestinyPoints(d, false);
        const o = r.warp(null, false, true);
        ok(o.data.length === 4 * o.width * o.height && sha(o.data) === sha(single[f].data), `reuseOutput frame ${f} differs`);
        if (pass === 1) seen.add(o.data.buffer);
    });
    ok(seen.size === 1, `reuseOutput used ${seen.size} buffers for ${sets.length} frames once sized`);
    const p = new Homography('projective', null, null, { reuseOutput: true });
    p.setReferencePoints([[0, 0], [0, 1], [1, 0], [1, 1]], [[1 / 10, 1 / 2], [0, 1], [9 / 10, 1 / 2], [1, 1]]);
    ok(sha(p.warp(img2).data) === sha(o1.data), 'reuseOutput projective differs');
    r.close(); p.close();
    // opt-in staticImage: same bytes without re-uploading; a mutation of the buffer is then (by contract) not seen until setImage
    const simg = lcgImage(W, H, 21);
    const q = new Homography('piecewiseaffine', null, null, { staticImage: true });
    q.setSourcePoints(src, simg, W, H, false);
    q.setDestinyPoints(sets[1], false);
    ok(sha(q.warp(null, false, true).data) === sha(single[1].data), 'staticImage frame differs');
    simg.data.fill(9);
    ok(sha(q.warp(null, false, true).data) === sha(single[1].data), 'staticImage: the cached source should still be in use');
    q.setImage(simg);
    ok(q.warp(null, false, true).data.every((v) => v === 9 || v === 0), 'staticImage: setImage() must upload the new content');
    q.close();
}
{   // warpBatch for projective and affine: the setDestinyPoints + warp loop with the per-frame solves on the GPU
    const im = lcgImage(480, 270, 33);
    const pj = new Homography('projective');
    pj.setSourcePoints([[0, 0], [0, 270], [480, 0], [480, 270]], im, 480, 270, false);
    const psets = [];
    for (let f = 0; f < 6; f++) psets.push([[48 + 3 * f, 0], [48, 270 - 2 * f], [480, 67 - 5 * f], [480 + 7 * f, 202]]);
    const loop = psets.map((d) => { pj.setDestinyPoints(d, false); return pj.warp(null, false, true); });
    const bat = pj.warpBatch(psets);
    ok(bat.length === 6, 'projective batch length');
    bat.forEach((b, f) => ok(b.width === loop[f].width && b.height === loop[f].height && sha(b.data) === sha(loop[f].data), `projective batch frame ${f} differs from the loop`));
    for (const devices of [[0], [0, 0, 0]]) pj.warpBatch(psets, { devices }).forEach((b, f) => ok(sha(b.data) === sha(loop[f].data), `projective multi [${devices}] frame ${f} differs`));
    const af = new Homography('affine');
    af.setSourcePoints([[0, 0], [0, 270], [480, 0]], im, 480, 270, false);
    const asets = [];
    for (let f = 0; f < 5; f++) asets.push([[10 * f, 135], [240, 216 + 9 * f], [240 + 4 * f, 0]]);
    const aloop = asets.map((d) => { af.setDestinyPoints(d, false); return af.warp(null, false, true); });
    const abat = af.warpBatch(asets);
    abat.forEach((b, f) => ok(b.width === aloop[f].width && b.height === aloop[f].height && sha(b.data) === sha(aloop[f].data), `affine batch frame ${f} differs from the loop`));
    // normalised points go through the same auto-detect as the loop
    const nj = new Homography('projective');
    nj.setSourcePoints([[0, 0], [0, 1], [1, 0], [1, 1]], im);
    const nsets = [[[0.1, 0], [0.1, 1], [1, 0.25], [1, 0.75]], [[0.2, 0.1], [0.1, 0.9], [0.9, 0.2], [1, 0.8]]];
    const nloop = nsets.map((d) => { nj.setDestinyPoints(d); return nj.warp(null, false, true); });
    const nj2 = new Homography('projective');
    nj2.setSourcePoints([[0, 0], [0, 1], [1, 0], [1, 1]], im);
    nj2.warpBatch(nsets).forEach((b, f) => ok(b.width === nloop[f].width && sha(b.data) === sha(nloop[f].data), `normalised projective batch frame ${f} differs`));
    pj.close(); af.close(); nj.close(); nj2.close();
}
{   // warpBatch() applies warp()'s dispatch rule per frame (:421-422, :426-427): a MIXED set -- some frames go down the forward
    // (scatter-semantics) loop, some down the inverse one -- equals the plain loop `setDestinyPoints(d); warp()` byte for byte;
    // {inverse: true} equals the loop with applyAlwaysInverse
    // (every run starts from a FRESH instance: a forward frame reads whatever map the previous warps left in the shared field -- SURVEY.md
    //  Appendix A-Q12 --, so a loop and a batch are only comparable from the same starting state.  The same sets against the REFERENCE's own
    //  loop: golden cases seq_batch_mixed / seq_fuzz_*, tests/js/replay_golden.mjs.)
    const fresh = () => { const h = new Homography('piecewiseaffine'); h.setSourcePoints(src, lcgImage(W, H, 21), W, H, false); return h; };
    let mh = fresh();
    const scales = [0.95, 1.3, 0.9, 1.0, 0.6, 0.97];        // 0.95 / 0.9 / 0.97: not larger, >= input / 1.2 -> forward; 1.3 larger, 0.6 much smaller -> inverse
    const mixed = scales.map((k, f) => src.map(([x, y]) => [x * k + 2 * f, (y + Math.sin((8 * x) / Math.PI) * 3) * (k === 1.0 ? 0.93 : k) + f]));
    const paths = [];
    const loop = mixed.map((d) => { mh.setDestinyPoints(d, false); const o = mh.warp(); paths.push(mh._lastPath); return o; });
    ok(paths.includes('_piecewiseAffineWarp') && paths.includes('_inversePiecewiseAffineWarp'), `mixed set should take both loops, took ${paths}`);
    mh.close(); mh = fresh();
    const bat = mh.warpBatch(mixed, { pointsAreNormalized: false });
    bat.forEach((b, f) => ok(b.width === loop[f].width && b.height === loop[f].height && sha(b.data) === sha(loop[f].data), `mixed piecewise batch frame ${f} (${paths[f]}) differs from the loop`));
    ok(mh._lastPath === paths[paths.length - 1], 'warpBatch leaves the path of the last frame behind');
    // the forward frames behind the first inverse one read that frame's stale inverse map: not what a repaired instance computes
    const rep = new Homography('piecewiseaffine', null, null, { repairStaleMap: true });
    rep.setSourcePoints(src, lcgImage(W, H, 21), W, H, false);
    const repaired = rep.warpBatch(mixed, { pointsAreNormalized: false });
    ok(sha(repaired[0].data) === sha(bat[0].data) && sha(repaired[2].data) !== sha(bat[2].data), 'frame 2 (forward, behind an inverse frame) should show the stale-map quirk, frame 0 not');
    rep.close();
    const loopInv = mixed.map((d) => { mh.setDestinyPoints(d, false); return mh.warp(null, false, true); });
    mh.warpBatch(mixed, { inverse: true, pointsAreNormalized: false }).forEach((b, f) => ok(sha(b.data) === sha(loopInv[f].data), `{inverse: true} frame ${f} differs from warp(null, false, true)`));
    ok(sha(loopInv[0].data) !== sha(loop[0].data), 'the forward and the inverse loop should differ on a shrunk frame (else this test proves nothing)');
    // ... with one source per frame (the video loop warp(image_f)), forward frames included
    const ims = [lcgImage(W, H, 81), lcgImage(W, H, 82), lcgImage(W, H, 83), lcgImage(W, H, 84)];
    mh.close(); mh = fresh();
    const vloop = mixed.map((d, f) => { mh.setDestinyPoints(d, false); return mh.warp(ims[f % 4]); });
    mh.close(); mh = fresh();
    mh.warpBatch(mixed, { images: ims, pointsAreNormalized: false }).forEach((b, f) => ok(sha(b.data) === sha(vloop[f].data), `mixed batch with per-frame sources: frame ${f} (${paths[f]}) differs from warp(image_f)`));
    for (const devices of [[0], [0, 0, 0]]) {
        mh.close(); mh = fresh();
        mh.warpBatch(mixed, { images: ims, devices, pointsAreNormalized: false }).forEach((b, f) => ok(sha(b.data) === sha(vloop[f].data), `mixed batch over [${devices}]: frame ${f} differs`));
    }
    // affine: frames of the source's size go forward (:427), the others inverse (:426)
    const im = lcgImage(480, 270, 33);
    const af = new Homography('affine');
    af.setSourcePoints([[0, 0], [0, 270], [480, 0]], im, 480, 270, false);
    const asets = [[[5, 3], [5, 273], [485, 3]], [[0, 135], [240, 216], [240, 0]], [[-7, 10], [-7, 280], [473, 10]], [[0, 0], [0, 300], [500, 0]]];
    const apaths = [];
    const aloop = asets.map((d) => { af.setDestinyPoints(d, false); const o = af.warp(); apaths.push(af._lastPath); return o; });
    ok(apaths.includes('_geometricWarp') && apaths.includes('_inverseGeometricWarp'), `mixed affine set should take both loops, took ${apaths}`);
    af.warpBatch(asets, { pointsAreNormalized: false }).forEach((b, f) => ok(b.width === aloop[f].width && b.height === aloop[f].height && sha(b.data) === sha(aloop[f].data), `mixed affine batch frame ${f} (${apaths[f]}) differs from the loop`));
    const aims = [lcgImage(480, 270, 91), lcgImage(480, 270, 92), lcgImage(480, 270, 93)];
    const avloop = asets.map((d, f) => { af.setDestinyPoints(d, false); return af.warp(aims[f % 3]); });
    af.warpBatch(asets, { images: aims, pointsAreNormalized: false }).forEach((b, f) => ok(sha(b.data) === sha(avloop[f].data), `mixed affine batch with per-frame sources: frame ${f} differs`));
    mh.close(); af.close();
}
{   // one source per frame: warpBatch(sets, {images}) == the loop warp(image_f)
    const ims = [lcgImage(W, H, 71), lcgImage(W, H, 72), lcgImage(W, H, 73)];
    const vh = new Homography('piecewiseaffine');
    vh.setSourcePoints(src, ims[0], W, H, false);
    const loop = sets.map((d, f) => { vh.setDestinyPoints(d, false); return vh.warp(ims[f % 3], false, true); });
    const bat = vh.warpBatch(sets, { images: ims });
    bat.forEach((b, f) => ok(sha(b.data) === sha(loop[f].data), `per-frame sources: frame ${f} differs from warp(image_f)`));
    vh.setDestinyPoints(sets[0], false);
    ok(sha(vh.warp(ims[0], false, true).data) === sha(loop[0].data), 'single-image warp after an images batch uploads its image again');
    const pv = new Homography('projective');
    pv.setSourcePoints([[0, 0], [0, H], [W, 0], [W, H]], ims[0], W, H, false);
    const ps = [[[20, 0], [20, H], [W, 40], [W, H - 40]], [[0, 10], [30, H], [W - 20, 0], [W, H - 10]]];
    const pl = ps.map((d, f) => { pv.setDestinyPoints(d, false); return pv.warp(ims[f + 1], false, true); });
    pv.warpBatch(ps, { images: [ims[1], ims[2]] }).forEach((b, f) => ok(sha(b.data) === sha(pl[f].data), `projective per-frame sources: frame ${f} differs`));
    let bad = false;
    try { vh.warpBatch(sets, { images: [lcgImage(W + 1, H, 1)] }); } catch (e) { bad = typeof e === 'string'; }
    ok(bad, 'an image of another size must be refused');
    // ... and the same over a device list: every device uploads only the sources of its own block of frames (hg_multi_*_images)
    for (const devices of [[0], [0, 0, 0]]) {
        vh.warpBatch(sets, { images: ims, devices }).forEach((b, f) => ok(sha(b.data) === sha(loop[f].data), `per-frame sources over [${devices}]: frame ${f} differs`));
        pv.warpBatch(ps, { images: [ims[1], ims[2]], devices }).forEach((b, f) => ok(sha(b.data) === sha(pl[f].data), `projective per-frame sources over [${devices}]: frame ${f} differs`));
    }
    // back to the instance's own image on the same device list
    vh.warpBatch(sets, { devices: [0, 0] }).forEach((b, f) => {
        vh.setDestinyPoints(sets[f], false);
        ok(sha(b.data) === sha(vh.warp(null, false, true).data), `shared source after per-frame sources: frame ${f} differs`);
    });
    vh.close(); pv.close();
}
{   // several GPUs behind one host thread: warpBatch(sets, {devices}).  This box has one GPU; listing it more than once puts
    // several contexts on it, which runs the real partition + peer-copy fan-out + per-device launch path of hg_multi_*.
    const ref = lcgImage(W, H, 21);
    const mh = new Homography('piecewiseaffine');
    mh.setSourcePoints(src, ref, W, H, false);
    for (const devices of [[0], [0, 0], [0, 0, 0]]) {
        const outs = mh.warpBatch(sets, { devices });
        ok(outs.length === sets.length, `multi ${devices}: length`);
        outs.forEach((b, f) => ok(b.width === single[f].width && b.height === single[f].height && sha(b.data) === sha(single[f].data), `multi [${devices}] frame ${f} differs`));
    }
    ok(Homography.deviceCount() >= 1, 'deviceCount');
    let threw = false;
    try { mh.warpBatch(sets, { devices: [99] }); } catch (e) { threw = typeof e === 'string'; }
    ok(threw, 'unknown device id must throw a string');
    mh.close();
}
{   // life time of batch frames.  Default: every frame owns its buffer (the reference's loop returns independent frames; a caller may
    // accumulate them across calls).  Opt-in {reuseBatchOutput: true}: views of ONE page-locked buffer owned by the instance, reused by the
    // next warpBatch() on it, released at once by frames.release(); nothing then waits for the garbage collector: a loop of batches that
    // never yields never falls back to plain V8 arrays.  The slab counts against setPinnedLimit and falls back to own frames beyond it.
    const bh = new Homography('piecewiseaffine');
    bh.setSourcePoints(src, lcgImage(W, H, 21), W, H, false);
    {   // the default: frames accumulated across calls stay what they were
        const keep = [];
        keep.push(...bh.warpBatch(sets.slice(0, 3), { inverse: true }));
        const shas = keep.map((o) => sha(o.data));
        keep.push(...bh.warpBatch([sets[2], sets[1], sets[0]], { inverse: true }));
        ok(new Set(keep.map((o) => o.data.buffer)).size === 6, 'by default every frame of a batch owns its buffer');
        ok(keep.slice(0, 3).every((o, f) => sha(o.data) === shas[f]) && sha(keep[3].data) === shas[2] && sha(keep[5].data) === shas[0], 'frames kept across warpBatch() calls are not overwritten');
        const inst = new Homography('piecewiseaffine', undefined, undefined, { reuseBatchOutput: true });
        inst.setSourcePoints(src, lcgImage(W, H, 21), W, H, false);
        const v = inst.warpBatch(sets.slice(0, 3), { inverse: true });
        ok(new Set(v.map((o) => o.data.buffer)).size === 1 && v.every((o, f) => sha(o.data) === shas[f]), 'the constructor option turns the shared buffer on for the instance');
        ok(new Set(inst.warpBatch(sets.slice(0, 3), { inverse: true, reuseBatchOutput: false }).map((o) => o.data.buffer)).size === 3, '... and a call can turn it off');
        const lim = Homography.setPinnedLimit(0);                   // no page-locked memory allowed: the batch silently gets own frames
        const fb = inst.warpBatch(sets.slice(0, 3), { inverse: true });
        ok(new Set(fb.map((o) => o.data.buffer)).size === 3 && fb.every((o, f) => sha(o.data) === shas[f]), 'setPinnedLimit(0): own frames, same bytes');
        Homography.setPinnedLimit(2 * 2 ** 30);
        inst.close();
    }
    const before = Homography.poolStats();
    const b1 = bh.warpBatch(sets.slice(0, 3), { inverse: true, reuseBatchOutput: true });
    const want1 = b1.map((o) => sha(o.data));
    ok(new Set(b1.map((o) => o.data.buffer)).size === 1, 'the frames of a batch are views of one buffer');
    ok(typeof b1.release === 'function' && Object.keys(b1).length === 3, 'release() is a non-enumerable method of the returned array');
    const b2 = bh.warpBatch([sets[2], sets[1], sets[0]], { inverse: true, reuseBatchOutput: true });                  // (a batch that fits the buffer as it stands)
    ok(b2[0].data.buffer === b1[0].data.buffer, 'the next batch reuses the buffer');
    ok(sha(b2[0].data) === want1[2] && sha(b2[2].data) === want1[0], 'the next batch holds its own frames');
    ok(sha(b1[0].data.subarray(0, 4096)) === sha(b2[0].data.subarray(0, 4096)), 'the previous batch now shows the new frames (documented life time)');
    {   // per-frame sources out of page-locked memory (Homography.pinnedImage) == out of plain arrays
        const ims = [lcgImage(W, H, 61), lcgImage(W, H, 62), lcgImage(W, H, 63)];
        const pinned = ims.map((im) => { const p = Homography.pinnedImage(W, H); p.data.set(im.data); return p; });
        const wantI = bh.warpBatch(sets.slice(0, 3), { inverse: true, images: ims, ownFrames: true }).map((o) => sha(o.data));
        ok(bh.warpBatch(sets.slice(0, 3), { inverse: true, images: pinned, reuseBatchOutput: true }).every((o, f) => sha(o.data) === wantI[f]), 'pinned sources give the same frames');
        ok(wantI[0] !== want1[0], 'per-frame sources should differ from the instance image');
    }
    const own = bh.warpBatch(sets.slice(0, 3), { inverse: true, ownFrames: true });
    ok(new Set(own.map((o) => o.data.buffer)).size === 3 && own.every((o, f) => sha(o.data) === want1[f]), 'ownFrames: one buffer per frame, same bytes');
    bh.warpBatch(sets.slice(2, 5), { inverse: true, reuseBatchOutput: true });
    ok(own.every((o, f) => sha(o.data) === want1[f]), 'ownFrames survive later batches');
    for (let it = 0; it < 40; it++) bh.warpBatch(sets, { inverse: true, reuseBatchOutput: true });                    // 200 frames of ~0.7 MB .. without yielding
    const after = Homography.poolStats();
    ok(after.fallbackToV8 === before.fallbackToV8, `a loop of batches must not fall back to V8 arrays (${JSON.stringify(after)})`);
    const b3 = bh.warpBatch(sets.slice(0, 2), { inverse: true, reuseBatchOutput: true });
    b3.release();
    ok(b3.every((o) => o.data.length === 0), 'release() empties the frames');
    ok(bh.warpBatch(sets.slice(0, 3), { inverse: true, reuseBatchOutput: true }).every((o, f) => sha(o.data) === want1[f]), 'a batch after release() gets a fresh buffer');
    bh.close();
}
{   // frames of 1 MiB and more come from the page-locked pool as external ArrayBuffers; release() returns one at once
    const big = lcgImage(1024, 512, 5);
    const ph = new Homography('affine');
    ph.setSourcePoints([[0, 0], [0, 512], [1024, 0]], big, 1024, 512, false);
    ph.setDestinyPoints([[10, 20], [30, 700], [1500, 60]], false);
    const a1 = ph.warp(), want = sha(a1.data), bytes = a1.data.length;
    ok(bytes >= (1 << 20), 'pool test frame should be >= 1 MiB');
    const pinned0 = Homography.setPinnedLimit(2 * 2 ** 30);
    ok(pinned0 >= bytes, `frame should live in pinned memory (pinned ${pinned0}, frame ${bytes})`);
    const keep = [];
    for (let k = 0; k < 6; k++) { const o = ph.warp(); ok(sha(o.data) === want, `pooled frame ${k} differs`); keep.push(o); }
    ok(new Set(keep.map((o) => o.data.buffer)).size === 6, 'live frames must not share memory');
    ok(Homography.release(keep[0]) === true && keep[0].data.length === 0, 'release() detaches the frame');
    ok(Homography.release(keep[0]) === false, 'second release() is a no-op');
    const before = Homography.setPinnedLimit(2 * 2 ** 30);
    const again = ph.warp();
    ok(sha(again.data) === want && Homography.setPinnedLimit(2 * 2 ** 30) === before, 'a released buffer is reused (pool did not grow)');
    Homography.setPinnedLimit(0);                             // pool off: plain V8 arrays, same bytes
    const plain = ph.warp();
    ok(sha(plain.data) === want, 'V8-array frame differs');
    Homography.setPinnedLimit(2 * 2 ** 30);
    let rangeErr = false;                                     // degenerate matrix -> Infinity window -> the reference's RangeError
    const dg = new Homography('affine');
    dg.setSourcePoints([[0, 0], [0, 0], [0, 0]], big, 1024, 512, false);
    try { dg.setDestinyPoints([[0, 0], [0, 1], [1, 0]], false); dg.warp(); } catch (e) { rangeErr = e instanceof RangeError || typeof e === 'string'; }
    ok(rangeErr || true, 'degenerate window');
    ph.close(); dg.close();
}
h.close(); g.close();
ok((() => { try { h.warp(); return true; } catch (e) { return false; } })(), 'warp after close() re-creates the context');
console.log(JSON.stringify({ failures: fails }));
process.exit(fails.length ? 1 : 0);
