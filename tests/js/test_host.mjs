// Host-side (no GPU) checks of the drop-in class's JavaScript: own Delaunay triangulator, error behaviour, CSS export.
import { triangulate, gridTriangles } from '../../homography.js_amd/js/delaunay.mjs';
import { Homography } from '../../homography.js_amd/js/Homography.mjs';
import { createRequire } from 'module';

const fails = [];
const ok = (c, m) => { if (!c) fails.push(m); };
function rng(seed) { let s = seed >>> 0 || 1; return () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s / 4294967296; }; }
function hullArea(P) {
    const p = P.map((q) => q.slice()).sort((a, b) => a[0] - b[0] || a[1] - b[1]);
    const cross = (o, a, b) => (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0]);
    const lo = [], up = [];
    for (const q of p) { while (lo.length >= 2 && cross(lo[lo.length - 2], lo[lo.length - 1], q) <= 0) lo.pop(); lo.push(q); }
    for (const q of p.reverse()) { while (up.length >= 2 && cross(up[up.length - 2], up[up.length - 1], q) <= 0) up.pop(); up.push(q); }
    const h = lo.slice(0, -1).concat(up.slice(0, -1));
    let a = 0; for (let i = 0; i < h.length; i++) { const j = (i + 1) % h.length; a += h[i][0] * h[j][1] - h[j][0] * h[i][1]; }
    return Math.abs(a) / 2;
}
function checkDelaunay(P, tag) {
    const t = triangulate(P);
    ok(t instanceof Uint32Array && t.length % 3 === 0 && t.length > 0, `${tag}: output type`);
    let area = 0;
    for (let i = 0; i < t.length; i += 3) {
        const [a, b, c] = [P[t[i]], P[t[i + 1]], P[t[i + 2]]];
        const d = 2 * (a[0] * (b[1] - c[1]) + b[0] * (c[1] - a[1]) + c[0] * (a[1] - b[1]));
        area += Math.abs(d) / 4;
        if (Math.abs(d) < 1e-12) { fails.push(`${tag}: degenerate triangle`); continue; }
        const ux = ((a[0] ** 2 + a[1] ** 2) * (b[1] - c[1]) + (b[0] ** 2 + b[1] ** 2) * (c[1] - a[1]) + (c[0] ** 2 + c[1] ** 2) * (a[1] - b[1])) / d;
        const uy = ((a[0] ** 2 + a[1] ** 2) * (c[0] - b[0]) + (b[0] ** 2 + b[1] ** 2) * (a[0] - c[0]) + (c[0] ** 2 + c[1] ** 2) * (b[0] - a[0])) / d;
        const r2 = (a[0] - ux) ** 2 + (a[1] - uy) ** 2;
        for (let q = 0; q < P.length; q++) {
            if (q === t[i] || q === t[i + 1] || q === t[i + 2]) continue;
            if ((P[q][0] - ux) ** 2 + (P[q][1] - uy) ** 2 < r2 * (1 - 1e-9)) { fails.push(`${tag}: point ${q} inside circumcircle of triangle ${i / 3}`); break; }
        }
    }
    const ha = hullArea(P);
    ok(Math.abs(area - ha) <= 1e-6 * ha, `${tag}: triangles cover ${area}, hull is ${ha}`);
    return t;
}
for (let k = 0; k < 12; k++) {
    const r = rng(900 + k), n = 5 + Math.floor(r() * 120), P = [];
    for (let i = 0; i < n; i++) P.push([r() * 1000, r() * 700]);
    checkDelaunay(P, `random${k}`);
}
{   // the 68-landmark style layout and a regular grid (co-circular quads)
    const G = []; for (let j = 0; j <= 5; j++) for (let i = 0; i <= 7; i++) G.push([i * 50, j * 40]);
    const t = checkDelaunay(G, 'grid');
    ok(t.length / 3 === 2 * 7 * 5, `grid: ${t.length / 3} triangles`);
    ok(gridTriangles(7, 5).length === 7 * 5 * 6, 'gridTriangles length');
    ok(triangulate(Float32Array.from([0, 0, 10, 0, 0, 10])).length === 3, 'typed-array input');
    ok(triangulate([[0, 0], [1, 1]]).length === 0, 'fewer than 3 points');
}
{   // the C-ABI triangulator (hg_triangulate, used by non-JS hosts) returns the same list, order included
    const hg = createRequire(import.meta.url)('../../homography.js_amd/lib/hgwarp.node');
    const r = rng(4242);
    let same = true;
    for (let k = 0; k < 60 && same; k++) {
        const n = 3 + Math.floor(r() * 150), p = new Float32Array(2 * n);
        for (let i = 0; i < 2 * n; i++) p[i] = k % 3 ? r() * 900 : Math.floor(r() * 9) * 25;      // every third set: lattice with duplicates
        const a = triangulate(p), b = hg.triangulate(p);
        same = a.length === b.length && a.every((v, i) => v === b[i]);
    }
    ok(same, 'hg_triangulate differs from js/delaunay.mjs');
}
{   // (the API's bare-string errors and the CSS export are pinned to the REFERENCE's own strings: tests/golden cases `errors_bare_strings`,
    //  `css_*`, replayed by tests/js/replay_golden.mjs; here only what has no counterpart in the reference)
    const expectThrow = (fn, part, tag) => { try { fn(); fails.push(`${tag}: did not throw`); } catch (e) { ok(typeof e === 'string' && e.includes(part), `${tag}: threw ${typeof e} ${e}`); } };
    expectThrow(() => new Homography('piecewiseaffine').setImage({ width: 4, height: 4 }), 'ImageData-shaped', 'HTMLImageElement-like input');
    expectThrow(() => new Homography().warp({ data: new Uint8ClampedArray(16), width: 2, height: 2 }, true), 'browser DOM', 'asHTMLPromise');
}
{   // life cycle of the page-locked frame pool, without a GPU (_poolTestFrames: the allocation path of warp() on malloc memory):
    // a loop that never yields to the event loop must not grow the pool without bound (weak references + requested collections),
    // frames that are alive never share memory, release() detaches, limit 0 means plain V8 arrays
    const B = 8 * 1024 * 1024;
    const require = createRequire(import.meta.url), hg = require('../../homography.js_amd/lib/hgwarp.node');
    const room = (n) => { if (hg.poolPressure(B, n)) { const v8 = require('v8'), vm = require('vm'); v8.setFlagsFromString('--expose-gc'); vm.runInNewContext('gc')(); hg.poolCollected(); } };
    let sum = 0;
    for (let it = 0; it < 400; it++) { room(1); const o = hg._poolTestFrames(1, B)[0]; ok(o.length === B, 'frame size'); sum += o[0]; }
    for (let it = 0; it < 60; it++) { room(8); for (const o of hg._poolTestFrames(8, B)) sum += o.length; }
    let st = hg.poolStats();
    ok(st.fallbackToV8 === 0 && st.buffers <= 80, `pool should stay bounded in a loop that never yields: ${JSON.stringify(st)}`);
    const live = hg._poolTestFrames(6, B);
    ok(new Set(live.map((o) => o.buffer)).size === 6, 'live frames share memory');
    live.forEach((o, k) => { o.fill(k + 1); });
    ok(live.every((o, k) => o[0] === k + 1 && o[B - 1] === k + 1), 'live frames overlap');
    ok(hg.release(live[2]) === true && live[2].length === 0 && hg.release(live[2]) === false, 'release');
    hg.setPinnedLimit(0);
    const before = hg.poolStats().allocated;
    const plain = hg._poolTestFrames(2, B);
    ok(plain[0].length === B && hg.poolStats().allocated === before, 'limit 0: plain V8 arrays');
    hg.setPinnedLimit(2 * 2 ** 30);
    ok(hg._poolTestFrames(1, 1000)[0].length === 1000, 'small frames are plain arrays');
}
console.log(JSON.stringify({ failures: fails }));
process.exit(fails.length ? 1 : 0);
