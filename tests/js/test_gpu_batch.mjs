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
h.close(); g.close();
ok((() => { try { h.warp(); return true; } catch (e) { return false; } })(), 'warp after close() re-creates the context');
console.log(JSON.stringify({ failures: fails }));
process.exit(fails.length ? 1 : 0);
