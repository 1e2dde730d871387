// The reference README's benchmark loop (`setDestinyPoints(dst); warp()` on a 400x400 source) through the drop-in JS class,
// end to end (host buffers in and out, fresh output array per call like the reference).   node tools/bench_node_readme.mjs
import { Homography } from '../homography.js_amd/js/Homography.mjs';
import { gridTriangles } from '../homography.js_amd/js/delaunay.mjs';
const W = 400, H = 400;
const data = new Uint8ClampedArray(W * H * 4);
{ let s = 1; for (let i = 0; i < data.length; i++) { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; data[i] = s >>> 24; } }
const img = { data, width: W, height: H };
const now = () => Number(process.hrtime.bigint()) / 1e6;
const res = {};
function loop(name, h, dsts, inverse) {
    h.setDestinyPoints(dsts[0], false);
    let out = h.warp(img, false, inverse);
    const t0 = now(); let n = 0;
    while (now() - t0 < 1500) { h.setDestinyPoints(dsts[n % dsts.length], false); out = h.warp(null, false, inverse); n++; }
    res[name] = { ms_per_frame: +((now() - t0) / n).toFixed(4), out: `${out.width}x${out.height}` };
}
for (const size of [200, 400, 800]) {
    const s = size / 400;
    const a = new Homography('affine');
    a.setSourcePoints([[0, 0], [0, H], [W, 0]], img, W, H, false);
    loop(`affine ${size}`, a, [[[0, 0], [4 * s, H * s], [W * s, 6 * s]], [[0, 0], [5 * s, H * s], [W * s, 7 * s]]], true);
    a.close();
    const p = new Homography('projective');
    p.setSourcePoints([[0, 0], [0, H], [W, 0], [W, H]], img, W, H, false);
    loop(`projective ${size}`, p, [[[0, 0], [10 * s, H * s], [W * s, 12 * s], [W * s * 0.95, H * s * 0.97]], [[0, 0], [9 * s, H * s], [W * s, 11 * s], [W * s * 0.96, H * s * 0.98]]], true);
    p.close();
    for (const [gx, gy] of [[1, 1], [18, 10], [107, 107]]) {
        const src = [], dsts = [[], []];
        for (let j = 0; j <= gy; j++) for (let i = 0; i <= gx; i++) {
            const x = i * (W / gx), y = j * (H / gy);
            src.push([x, y]);
            for (let f = 0; f < 2; f++) dsts[f].push([x * s, (0.02 * H / gy + y + Math.sin(((8 + f) * x) / Math.PI) * 0.02 * H / gy) * s]);
        }
        const h = new Homography('piecewiseaffine');
        Homography.triangulate = () => gridTriangles(gx, gy);
        h.setSourcePoints(src, img, W, H, false);
        loop(`piecewise ${2 * gx * gy} tri ${size}`, h, dsts.map((d) => Float32Array.from(d.flat())), true);     // (typed-array point sets: no per-frame flattening)
        h.close();
    }
}
console.log(JSON.stringify(res));
