// End-to-end timing of the drop-in JS class (host buffers in, host buffers out: PCIe + allocation included), C3 workload.
//   node tools/bench_node.mjs [seconds]
import { Homography } from '../homography.js_amd/js/Homography.mjs';
import { gridTriangles } from '../homography.js_amd/js/delaunay.mjs';

const W = 3840, H = 2160, nx = 10, ny = 10, A = 40, budget = Number(process.argv[2] || 3);
const data = new Uint8ClampedArray(W * H * 4);
{ let s = 1; for (let i = 0; i < data.length; i++) { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; data[i] = s >>> 24; } }
const src = [], dsts = [];
for (let j = 0; j <= ny; j++) for (let i = 0; i <= nx; i++) src.push([i * (W / nx), j * (H / ny)]);
for (let n = 8; n < 12; n++) dsts.push(src.map(([x, y]) => [x, A + y + Math.sin((n * x) / Math.PI) * A]));
const now = () => Number(process.hrtime.bigint()) / 1e6;

const h = new Homography('piecewiseaffine');
h.setSourcePoints(src, { data, width: W, height: H }, W, H, false);
h.setTriangles(gridTriangles(nx, ny));
h.setDestinyPoints(dsts[0], false);
let out = h.warp();                                             // first call: context creation, buffers
const res = { workload: `C3 ${W}x${H}, 200 triangles`, node: process.version };
// between the blocks: a few event-loop turns (+ a collection when run with --expose-gc) so that frames of the previous block
// are finalized and their pooled buffers come back (Node 12 runs N-API finalizers from the event loop only)
const snap = () => { const p = Homography.poolStats(); return `${p.buffers} buffers/${p.inUse} in use, reused ${p.reused}, new ${p.allocated}, v8 ${p.fallbackToV8}, reaped ${p.reapedByWeakRef}, forced gc ${p.forcedCollections}, finalized ${p.finalized}`; };
const settle = async () => { out = null; for (let k = 0; k < 4; k++) { if (global.gc) global.gc(); await new Promise((r) => setImmediate(r)); } };
(async () => {
{   // plain V8 arrays only (pool off): what a fresh Uint8ClampedArray per frame costs by itself
    Homography.setPinnedLimit(0);
    let frames = 0, px = 0; const t0 = now();
    while (now() - t0 < budget * 1e3) { h.setDestinyPoints(dsts[frames % 4], false); out = h.warp(); frames++; px += out.width * out.height; }
    const ms = now() - t0;
    res.warp_loop_v8_arrays = { frames, ms_per_frame: +(ms / frames).toFixed(3), mpix_per_s: +(px / ms / 1e3).toFixed(1) , pool: snap() };
    Homography.setPinnedLimit(2 * 2 ** 30);
}
await settle();
{   // the reference's per-frame loop: setDestinyPoints(dst_f); warp()  (never yields: pooled buffers only come back when the pool cap is hit ... never)
    let frames = 0, px = 0; const t0 = now();
    while (now() - t0 < budget * 1e3) { h.setDestinyPoints(dsts[frames % 4], false); out = h.warp(); frames++; px += out.width * out.height; }
    const ms = now() - t0;
    res.warp_loop = { frames, ms_per_frame: +(ms / frames).toFixed(3), mpix_per_s: +(px / ms / 1e3).toFixed(1) , pool: snap() };
}
await settle();
{   // the same loop when the consumer hands each frame back before asking for the next (Homography.release): pooled pinned buffers
    let frames = 0, px = 0; const t0 = now();
    while (now() - t0 < budget * 1e3) { h.setDestinyPoints(dsts[frames % 4], false); out = h.warp(); frames++; px += out.width * out.height; Homography.release(out); }
    const ms = now() - t0;
    res.warp_loop_release = { frames, ms_per_frame: +(ms / frames).toFixed(3), mpix_per_s: +(px / ms / 1e3).toFixed(1) , pool: snap() };
}
await settle();
{   // the same loop with the opt-in single output buffer (reuseOutput)
    h.reuseOutput = true;
    let frames = 0, px = 0; const t0 = now();
    while (now() - t0 < budget * 1e3) { h.setDestinyPoints(dsts[frames % 4], false); out = h.warp(); frames++; px += out.width * out.height; }
    const ms = now() - t0;
    res.warp_loop_reuse = { frames, ms_per_frame: +(ms / frames).toFixed(3), mpix_per_s: +(px / ms / 1e3).toFixed(1) , pool: snap() };
}
{   // ... and a source the caller promises not to mutate (staticImage): no re-upload per warp either
    h.staticImage = true;
    let frames = 0, px = 0; const t0 = now();
    while (now() - t0 < budget * 1e3) { h.setDestinyPoints(dsts[frames % 4], false); out = h.warp(); frames++; px += out.width * out.height; }
    const ms = now() - t0;
    res.warp_loop_reuse_static = { frames, ms_per_frame: +(ms / frames).toFixed(3), mpix_per_s: +(px / ms / 1e3).toFixed(1) , pool: snap() };
    h.reuseOutput = false; h.staticImage = false;
}
await settle();
{   // the same frames as one GPU pass
    const F = 8, sets = Array.from({ length: F }, (_, f) => dsts[f % 4]);
    let frames = 0, px = 0; const t0 = now();
    while (now() - t0 < budget * 1e3) { const outs = h.warpBatch(sets, { reuseBatchOutput: true }); frames += F; for (const o of outs) px += o.width * o.height; }
    const ms = now() - t0;
    res.warp_batch8 = { frames, ms_per_frame: +(ms / frames).toFixed(3), mpix_per_s: +(px / ms / 1e3).toFixed(1) , pool: snap() };
}
await settle();
{   // one source per frame (the video loop warp(image_f)): uploads pipelined with the downloads, frames in the instance's slab
    const F = 8, sets = Array.from({ length: F }, (_, f) => dsts[f % 4]);
    const images = Array.from({ length: F }, (_, f) => ({ data: Uint8ClampedArray.from(data.subarray(0, data.length)), width: W, height: H }));
    images.forEach((im, f) => { im.data[0] = f; });
    h.warpBatch(sets, { images, reuseBatchOutput: true });
    let frames = 0, px = 0; const t0 = now();
    while (now() - t0 < budget * 1e3) { const outs = h.warpBatch(sets, { images, reuseBatchOutput: true }); frames += F; for (const o of outs) px += o.width * o.height; }
    const ms = now() - t0;
    res.warp_batch8_images = { frames, ms_per_frame: +(ms / frames).toFixed(3), mpix_per_s: +(px / ms / 1e3).toFixed(1) , pool: snap() };
    {   // ... with the sources in page-locked memory (Homography.pinnedImage): uploads are asynchronous DMA, the two directions really overlap
        const pinned = Array.from({ length: F }, (_, f) => { const im = Homography.pinnedImage(W, H); im.data.set(images[f].data); return im; });
        h.warpBatch(sets, { images: pinned, reuseBatchOutput: true });
        let fr = 0, p2 = 0; const t2 = now();
        while (now() - t2 < budget * 1e3) { const outs = h.warpBatch(sets, { images: pinned, reuseBatchOutput: true }); fr += F; for (const o of outs) p2 += o.width * o.height; }
        const ms2 = now() - t2;
        res.warp_batch8_images_pinned = { frames: fr, ms_per_frame: +(ms2 / fr).toFixed(3), mpix_per_s: +(p2 / ms2 / 1e3).toFixed(1) , pool: snap() };
    }
    // the same as the plain loop warp(image_f) with reuseOutput (upload, kernel, download one after the other)
    h.reuseOutput = true;
    frames = 0; px = 0; const t1 = now();
    while (now() - t1 < budget * 1e3) { h.setDestinyPoints(dsts[frames % 4], false); out = h.warp(images[frames % F]); frames++; px += out.width * out.height; }
    const ms1 = now() - t1;
    res.warp_loop_images_reuse = { frames, ms_per_frame: +(ms1 / frames).toFixed(3), mpix_per_s: +(px / ms1 / 1e3).toFixed(1) , pool: snap() };
    h.reuseOutput = false;
    h.setImage({ data, width: W, height: H });
}
await settle();
{   // the old life time of batch frames: every frame its own pooled buffer, back when V8 collects it ({ownFrames: true})
    const F = 8, sets = Array.from({ length: F }, (_, f) => dsts[f % 4]);
    let frames = 0, px = 0; const t0 = now();
    while (now() - t0 < budget * 1e3) { const outs = h.warpBatch(sets, { ownFrames: true }); frames += F; for (const o of outs) px += o.width * o.height; }
    const ms = now() - t0;
    res.warp_batch8_own_frames = { frames, ms_per_frame: +(ms / frames).toFixed(3), mpix_per_s: +(px / ms / 1e3).toFixed(1) , pool: snap() };
}
await settle();
{   // batch with release of every frame after use
    const F = 8, sets = Array.from({ length: F }, (_, f) => dsts[f % 4]);
    let frames = 0, px = 0; const t0 = now();
    while (now() - t0 < budget * 1e3) { const outs = h.warpBatch(sets, { ownFrames: true }); frames += F; for (const o of outs) { px += o.width * o.height; Homography.release(o); } }
    const ms = now() - t0;
    res.warp_batch8_release = { frames, ms_per_frame: +(ms / frames).toFixed(3), mpix_per_s: +(px / ms / 1e3).toFixed(1) , pool: snap() };
}
await settle();
res.pinned_bytes = Homography.setPinnedLimit(2 * 2 ** 30);
// the reference-style loop driven from the event loop (one frame per tick, like a requestAnimationFrame / stream consumer):
// Node runs the finalizers of collected frames between ticks, so pooled buffers come back without release()
{
    let frames = 0, px = 0; const t0 = now();
    while (now() - t0 < budget * 1e3) {
        h.setDestinyPoints(dsts[frames % 4], false); out = h.warp(); frames++; px += out.width * out.height;
        await new Promise((r) => setImmediate(r));
    }
    const ms = now() - t0;
    res.warp_loop_async_tick = { frames, ms_per_frame: +(ms / frames).toFixed(3), mpix_per_s: +(px / ms / 1e3).toFixed(1), pinned_bytes: Homography.setPinnedLimit(2 * 2 ** 30) , pool: snap() };
}
h.close();
console.log(JSON.stringify(res));
})();
