// Random API call sequences for the Homography class, as plain data (both sides of a comparison build their own arguments from them):
// used by tests/js/fuzz_ref_sequences.mjs (live differential fuzz, build container) and tests/golden/gen_golden.mjs (the committed
// `seq_fuzz_*` golden cases).  The mix is tuned around what the reference keeps cached across calls (SURVEY.md Appendix A-Q12): warps
// that flip between the forward and the inverse loop (scales around the thresholds of warp() :421), source points / triangles / images
// re-set in the middle, normalised and pixel coordinates, typed-array arguments (aliased and mutated in place by the class).
export function rng(seed) { let s = (seed >>> 0) || 1; return () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s / 4294967296; }; }
export function lcgImage(w, h, seed) {
    const data = new Uint8ClampedArray(w * h * 4);
    let s = seed >>> 0;
    for (let i = 0; i < data.length; i++) { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; data[i] = s >>> 24; }
    return { data, width: w, height: h };
}

/** opt.triangles(srcPoints, grid) -> flat triangle list for setTriangles ops; opt.transform: fix the constructor's transform; opt.batches: false = no warpBatch ops. */
export function makeScript(r, opt = {}) {
    const pick = (a) => a[Math.floor(r() * a.length)];
    const W = 20 + Math.floor(r() * 50), H = 16 + Math.floor(r() * 44);
    const images = { a: { w: W, h: H, seed: 1 + Math.floor(r() * 1000) }, b: { w: W, h: H, seed: 1 + Math.floor(r() * 1000) },
                     c: { w: 16 + Math.floor(r() * 40), h: 16 + Math.floor(r() * 40), seed: 7 } };
    const transform = opt.transform || pick(['piecewiseaffine', 'piecewiseaffine', 'piecewiseaffine', 'piecewiseaffine', 'auto', 'auto', 'affine', 'projective']);
    let nPts;                                                // how many points the instance works with (fixed per script, like a real caller)
    const grid = { nx: 1 + Math.floor(r() * 4), ny: 1 + Math.floor(r() * 4) };
    if (transform === 'affine') nPts = 3; else if (transform === 'projective') nPts = 4;
    else if (transform === 'auto') nPts = pick([3, 4, (grid.nx + 1) * (grid.ny + 1)]);
    else nPts = (grid.nx + 1) * (grid.ny + 1);
    const srcPts = (norm, w, h) => {
        const p = [];
        if (nPts === 3) p.push([0, 0], [0, 1], [1, 0]);
        else if (nPts === 4 && transform !== 'piecewiseaffine') p.push([0, 0], [0, 1], [1, 0], [1, 1]);
        else for (let j = 0; j <= grid.ny; j++) for (let i = 0; i <= grid.nx; i++) p.push([i / grid.nx, j / grid.ny]);
        const inset = r() < 0.3 ? 0.1 + r() * 0.2 : 0, jit = r() < 0.4 ? r() * 0.15 : 0;
        return p.map(([x, y]) => { x = inset + x * (1 - 2 * inset) + (r() - 0.5) * jit / Math.max(grid.nx, 1); y = inset + y * (1 - 2 * inset) + (r() - 0.5) * jit / Math.max(grid.ny, 1);
                                   return norm ? [x, y] : [x * w, y * h]; });
    };
    const dstOf = (src, norm, w, h) => {
        // scale classes around warp()'s dispatch thresholds (:421): shrink a lot / a little / exact / grow
        const sc = () => pick([0.45 + r() * 0.3, 0.84 + r() * 0.16, 0.84 + r() * 0.16, 1, 1.02 + r() * 0.5]);
        const sx = sc(), sy = r() < 0.6 ? sx : sc(), ox = r() < 0.5 ? 0 : (r() - 0.3) * 0.4, oy = r() < 0.5 ? 0 : (r() - 0.3) * 0.4, jit = r() < 0.5 ? r() * 0.08 : 0;
        return src.map(([x, y]) => { const u = norm ? 1 : w, v = norm ? 1 : h;
                                     return [(x / u * sx + ox + (r() - 0.5) * jit) * (norm ? 1 : w), (y / v * sy + oy + (r() - 0.5) * jit) * (norm ? 1 : h)]; });
    };
    const ptsArg = (p) => (r() < 0.3 ? { f32: p.flat() } : p);
    const ctorSize = r() < 0.3;
    const script = [['new', transform, ctorSize ? W : null, ctorSize ? H : null]];
    let lastSrc = null, lastNorm = true;
    const genSrc = () => { lastNorm = r() < 0.5; lastSrc = srcPts(lastNorm, W, H); return lastSrc; };
    const imgKey = () => pick(['a', 'a', 'b', null]);
    // opening: one of the usual set-up orders
    const open = Math.floor(r() * 4);
    if (open === 0) script.push(['setReferencePoints', ptsArg(genSrc()), ptsArg(dstOf(lastSrc, r() < 0.7 ? lastNorm : !lastNorm, W, H)), imgKey()]);
    else if (open === 1) script.push(['setSourcePoints', ptsArg(genSrc()), imgKey(), r() < 0.3 ? W : null, r() < 0.3 ? H : null], ['setDestinyPoints', ptsArg(dstOf(lastSrc, lastNorm, W, H))]);
    else if (open === 2) script.push(['setImage', 'a'], ['setSourcePoints', ptsArg(genSrc())], ['setDestinyPoints', ptsArg(dstOf(lastSrc, r() < 0.8 ? lastNorm : !lastNorm, W, H))]);
    else script.push(['setSourcePoints', ptsArg(genSrc())], ['setDestinyPoints', ptsArg(dstOf(lastSrc, lastNorm, W, H))], ['setImage', 'a']);
    const nOps = 3 + Math.floor(r() * 10);
    for (let k = 0; k < nOps; k++) {
        const x = r();
        if (x < 0.40) script.push(['warp', r() < 0.35 ? pick(['a', 'b']) : null, r() < 0.2]);
        else if (x < 0.68) script.push(['setDestinyPoints', ptsArg(dstOf(lastSrc, r() < 0.85 ? lastNorm : !lastNorm, W, H)), r() < 0.1 ? (r() < 0.5) : null]);
        else if (x < 0.76) script.push(['setSourcePoints', ptsArg(genSrc()), r() < 0.3 ? imgKey() : null, null, null, r() < 0.1 ? lastNorm : null]);
        else if (x < 0.82) script.push(['setImage', pick(['a', 'b', 'b', 'c'])]);
        else if (x < 0.88 && nPts > 4) {
            const t = opt.triangles(lastSrc, grid);
            const mode = Math.floor(r() * 3), T = t.length / 3;
            let tt = t;
            if (mode === 0 && T > 1) tt = t.slice(0, 3 * (1 + Math.floor(r() * (T - 1))));                  // fewer triangles
            else if (mode === 1) { tt = []; for (let i = T - 1; i >= 0; i--) tt.push(t[3 * i], t[3 * i + 1], t[3 * i + 2]); }   // reversed order
            else tt = t.concat(t.slice(0, 3));                                                                  // one more (a duplicate)
            script.push(['setTriangles', tt]);
        }
        else if (x < 0.93) script.push(['setReferencePoints', ptsArg(genSrc()), ptsArg(dstOf(lastSrc, lastNorm, W, H)), r() < 0.3 ? imgKey() : null]);
        else if (opt.batches !== false) {
            const sets = []; const n = 2 + Math.floor(r() * 3);
            for (let i = 0; i < n; i++) sets.push(dstOf(lastSrc, lastNorm, W, H));
            script.push(['warpBatch', sets, r() < 0.2]);
        }
    }
    if (!script.some((o) => o[0] === 'warp')) script.push(['warp', 'a', false]);
    return { images, script, grid };
}

