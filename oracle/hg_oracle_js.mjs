// hg_oracle_js.mjs -- JavaScript restatement of the reference's inverse piecewise-affine path, for ONE purpose: timing
// "the reference's algorithm under Node.js on this box's host cores" next to the GPU numbers (bench.py cpu_baseline.node).
//
// TEST INFRASTRUCTURE ONLY (like oracle/hg_oracle.c): never imported by the product.  It follows the reference's
// Homography.js (v1.8.0) function by function -- affineMatrixFromTriangles :1265-1306, inverseAffineMatrix :1345-1365,
// fillTriangle / defineTriangleLineEquations / predictXLimits :1111-1197, _buildInverseTrianglesCorrespondencesMatrix
// :845-861, _inversePiecewiseAffineWarp :1029-1058 -- in plain JS Numbers, so its speed is what the reference's own loops
// achieve under this Node version (single thread).  Pinned: `node oracle/hg_oracle_js.mjs check` compares it with the
// golden vectors generated from the reference itself (tests/golden).
//
//   node oracle/hg_oracle_js.mjs check                 -> {"checked": n, "failures": [...]}
//   node oracle/hg_oracle_js.mjs bench C3 12           -> times frames of the C3 workload for ~12 s, prints Mpixels/s
import fs from 'fs';
import path from 'path';
import crypto from 'crypto';
import { fileURLToPath } from 'url';

const HERE = path.dirname(fileURLToPath(import.meta.url));

import core from './hg_oracle_core.cjs';

const { warpInversePiecewise } = core;                                 // the loops themselves: oracle/hg_oracle_core.cjs
export { warpInversePiecewise };

// ---------------------------------------------------------------- workloads (same generators as homography.js_amd/workloads.py)
function lcgImage(w, h, seed) {
    const data = new Uint8ClampedArray(w * h * 4);
    let s = seed >>> 0;
    for (let i = 0; i < data.length; i++) { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; data[i] = s >>> 24; }
    return data;
}
function sinGrid(W, H, nx, ny, A, n) {
    const sp = [], dp = [], tris = [];
    for (let j = 0; j <= ny; j++) for (let i = 0; i <= nx; i++) { const x = i * (W / nx), y = j * (H / ny); sp.push(x, y); dp.push(x, A + y + Math.sin((n * x) / Math.PI) * A); }
    for (let j = 0; j < ny; j++) for (let i = 0; i < nx; i++) { const a = j * (nx + 1) + i; tris.push(a, a + 1, a + nx + 1, a + 1, a + nx + 2, a + nx + 1); }
    return { sp: Float32Array.from(sp), dp: Float32Array.from(dp), tris: Uint32Array.from(tris) };
}
const geomOf = (p) => { let a = Infinity, b = Infinity, c = -Infinity, d = -Infinity; for (let i = 0; i < p.length; i += 2) { a = Math.min(a, p[i]); c = Math.max(c, p[i]); b = Math.min(b, p[i + 1]); d = Math.max(d, p[i + 1]); }
    return [Math.round(a), Math.round(b), Math.round(c) - Math.round(a), Math.round(d) - Math.round(b)]; };
const CONFIGS = { C3: [3840, 2160, 10, 10, 40], C5: [7680, 4320, 50, 50, 80], small: [160, 96, 10, 10, 4] };

const mode = process.argv[2];
if (mode === 'bench') {
    const [W, H, nx, ny, A] = CONFIGS[process.argv[3] || 'C3'];
    const budget = Number(process.argv[4] || 12);
    const img = lcgImage(W, H, 1);
    let frames = 0, px = 0;
    const t0 = process.hrtime.bigint();
    while (frames === 0 || Number(process.hrtime.bigint() - t0) / 1e9 < budget) {
        const g = sinGrid(W, H, nx, ny, A, 8 + (frames % 4)), [xo, yo, ow, oh] = geomOf(g.dp);
        warpInversePiecewise(g.sp, g.dp, g.tris, img, W, H, 0, 0, xo, yo, ow, oh);
        frames++; px += ow * oh;
    }
    const s = Number(process.hrtime.bigint() - t0) / 1e9;
    console.log(JSON.stringify({ config: process.argv[3] || 'C3', frames, seconds: +s.toFixed(2), mpix_per_s: +(px / s / 1e6).toFixed(2), node: process.version }));
} else if (mode === 'check') {
    const G = JSON.parse(fs.readFileSync(path.join(HERE, '..', 'tests', 'golden', 'golden.json'), 'utf8'));
    const BL = fs.readFileSync(path.join(HERE, '..', 'tests', 'golden', 'golden_blobs.bin'));
    const view = (ref, C) => { const b = BL.slice(ref.off, ref.off + ref.len); return new C(b.buffer.slice(b.byteOffset, b.byteOffset + b.length)); };
    const f32 = (x) => new Float32Array((Array.isArray(x) ? Uint32Array.from(x) : view(x, Uint32Array)).buffer);
    const f64 = (xs) => { const d = new Float64Array(xs.length), u = new Uint32Array(d.buffer); xs.forEach((h, i) => { u[2 * i + 1] = parseInt(h.slice(0, 8), 16); u[2 * i] = parseInt(h.slice(8), 16); }); return d; };
    const sha = (t) => crypto.createHash('sha256').update(Buffer.from(t.buffer, t.byteOffset, t.byteLength)).digest('hex');
    const failures = [];
    let n = 0;
    for (const c of G.cases) {
        if (/_(4k|8k|1080p)/.test(c.name) || /int16_wrap/.test(c.name)) continue;
        if (/^(seq_|readme_|css_|errors_)/.test(c.name)) continue;   // call-sequence cases: replayed through the class over these loops (tests/js/replay_golden.mjs --dry)
        c.warps.forEach((w, k) => {
            const spec = Object.values(c.images)[0], img = lcgImage(spec.w, spec.h, spec.seed);
            if (w.transform !== 'piecewiseaffine') {                   // the two geometric loops, matrices as the reference resolved them
                const kind = w.transform === 'affine' ? 0 : 1, inverse = w.path === '_inverseGeometricWarp';
                const mat = (mm) => (mm.f32 ? f32(mm.f32) : f64(mm.f64));
                const out = inverse ? core.inverseGeometricLoop(kind, mat(w.invMatrix), img, w.W, w.H, w.xOff, w.yOff, w.objW, w.objH)
                                    : core.forwardGeometricLoop(kind, mat(w.matrix), img, w.W, w.H, w.xOff, w.yOff, w.objW, w.objH);
                n++;
                if (sha(out) !== w.out.sha) failures.push(`${c.name}#${k}: RGBA (${w.path})`);
                return;
            }
            const tris = c.triangles.u32blob ? view(c.triangles.u32blob, Uint32Array) : Uint32Array.from(c.triangles);
            if (w.path === '_piecewiseAffineWarp') {                   // forward loop over the forward map of the source points
                const sp = f32(w.srcPoints), mw = w.maxSrcX - w.minSrcX, mh = w.maxSrcY - w.minSrcY;
                const map = core.buildTriangleMap(sp, tris, mw, mh, w.minSrcY);
                const out = core.forwardPiecewiseLoop(core.piecewiseMatrices(sp, f32(w.dstPoints), tris), map, img, w.W, w.minSrcX, w.minSrcY, w.maxSrcX, w.maxSrcY, w.xOff, w.yOff, w.objW, w.objH);
                n++;
                if (sha(out) !== w.out.sha) failures.push(`${c.name}#${k}: RGBA (forward piecewise)`);
                if (sha(map) !== w.map.sha) failures.push(`${c.name}#${k}: forward map`);
                return;
            }
            const r = warpInversePiecewise(f32(w.srcPoints), f32(w.dstPoints), tris, img, w.W, w.H, w.minSrcX, w.minSrcY, w.xOff, w.yOff, w.objW, w.objH);
            n++;
            if (sha(r.out) !== w.out.sha) failures.push(`${c.name}#${k}: RGBA`);
            if (sha(r.map) !== w.map.sha) failures.push(`${c.name}#${k}: map`);
        });
    }
    console.log(JSON.stringify({ checked: n, failures }));
    process.exit(failures.length ? 1 : 0);
} else {
    console.error('usage: hg_oracle_js.mjs check | bench <C3|C5|small> [seconds]');
    process.exit(2);
}
