// Homography.mjs -- drop-in `Homography` class for Node.js whose per-pixel work runs on an AMD MI355X (gfx950).
//
// Same public surface and state machine as Eric-Canas/Homography.js v1.8.0 (`import {Homography} from ...`):
//     new Homography(transform = 'auto', width = null, height = null)
//     setReferencePoints / setSourcePoints / setDestinyPoints / setImage / setTriangles / warp
//     getTransformationMatrixAsCSS
// Images are ImageData-shaped `{data: Uint8ClampedArray, width, height}` (what the reference accepts in Node, :297-299)
// and warp() returns the same shape.  The pixel loops, the triangle map and the per-triangle solves are NOT done in
// JavaScript: they go through the N-API addon lib/hgwarp.node -> C ABI libhgwarp.so -> HIP kernels.  There is no CPU
// fallback: without the addon or without a gfx950 GPU, construction throws.
//
// file:line comments cite the reference's Homography.js for the behaviour each method reproduces.  Reference quirks
// that are observable through the API are kept (SURVEY.md Appendix A), e.g. normalisation auto-detect (> 8.0), in-place
// (de)normalisation of caller-owned typed arrays, re-solving the inverse from swapped point sets.
//
// Differences, all Node-only consequences of having no DOM:
//   * HTMLImageElement inputs / asHTMLPromise / transformHTMLElement throw (the reference needs a browser for them too);
//   * (the reference's stale-state behaviour IS reproduced -- SURVEY.md Appendix A-Q12: one field holds the forward and the inverse
//     triangle map (:819-820 vs :847-848), so a forward piecewise warp that follows an inverse one indexes the stale INVERSE map
//     (:957); the per-triangle matrices survive setSourcePoints / setTriangles until the next setDestinyPoints (:252-255, :519,
//     :766).  The class mirrors both caches as value snapshots (`_map`, `_pm`) and hands the GPU exactly what the reference's loops
//     would read.  `new Homography(t, w, h, {repairStaleMap: true})` opts out: forward warps then always read the forward map of the
//     current mesh and every warp the matrices of the current point sets -- what the author evidently meant);
//   * frames of 1 MiB and more live in pooled page-locked memory handed out as external ArrayBuffers (no behavioural
//     difference; `Homography.release(imageData)` optionally returns a frame to the pool at once);
//   * Delaunator is not bundled: `Homography.triangulate` (default: ./delaunay.mjs, a restatement of delaunator 5's
//     sweep-hull algorithm that follows it step for step, so that triangle order and the diagonals of cocircular quads --
//     which decide pixels where triangles meet -- come out the same) supplies the triangles.  delaunator@5.0.0's source is
//     absent from the reference tree and none of its tests pins the triangulation, so identity with it is NOT claimed as
//     tested ("triangulation parity unpinned"); for guaranteed identical meshes pass delaunator's own output through
//     setTriangles() or assign Homography.triangulate.
import { createRequire } from 'module';
import { fileURLToPath } from 'url';
import path from 'path';
import { triangulate as defaultTriangulate } from './delaunay.mjs';

const require = createRequire(import.meta.url);
const HERE = path.dirname(fileURLToPath(import.meta.url));

let native = null;
function addon() {
    if (native === null) {
        const p = process.env.HGWARP_ADDON || path.join(HERE, '..', 'lib', 'hgwarp.node');
        try { native = require(p); } catch (e) {
            throw (`hgwarp: cannot load the native addon ${p} (${e && e.message ? e.message : e}); build it with \`make -C homography.js_amd\` -- there is no CPU fallback`);
        }
    }
    return native;
}

// When the page-locked frame pool starves (hgwarp_napi.c: poolPressure) a full collection would hand its dead frames back at
// once.  The library only ever asks for one when the USER started node with --expose-gc (global.gc exists); it never switches
// V8 flags itself and never forces collections on a host application that did not opt in -- without the flag, frames simply
// fall back to plain V8 arrays until V8 collects by itself (or the caller uses Homography.release(frame) / reuseOutput).
function collectGarbage() {
    if (typeof global.gc !== 'function') return false;
    global.gc();
    return true;
}
function makeRoomFor(native, bytes, n) {
    if (bytes >= 1048576 && native.poolPressure(bytes, n) && collectGarbage()) native.poolCollected();
}

const TRANSFORMS = ['auto', 'piecewiseaffine', 'affine', 'projective'];
const CSS_DECIMALS = 5;                 // :31
const NORMALIZED_MAX = 8.0;             // :36  anything above is taken as pixel coordinates
const AFFINE = 0, PROJECTIVE = 1;

class WarpedImage {                     // ImageData-shaped result for runtimes without a global ImageData
    constructor(data, width, height) { this.data = data; this.width = width; this.height = height; }
}
const makeImageData = (data, w, h) => (typeof ImageData !== 'undefined' ? new ImageData(data, w, h) : new WarpedImage(data, w, h));

const anyAbove = (arr, limit) => { for (let i = 0; i < arr.length; i++) if (arr[i] > limit) return true; return false; };   // :1539
const toF32 = (points) => (ArrayBuffer.isView(points) ? points : new Float32Array(points.flat()));                          // :220 / :339
function scalePoints(p, sx, sy) { for (let i = 0; i < p.length; i++) p[i] = (i % 2) === 0 ? p[i] * sx : p[i] * sy; }       // :1603
function unscalePoints(p, sx, sy) { for (let i = 0; i < p.length; i++) p[i] = (i % 2) === 0 ? p[i] / sx : p[i] / sy; }     // :1621
// `new Uint8ClampedArray(n)` of the reference (:991, :1040) throws a RangeError for lengths V8 cannot allocate (Infinity from a
// degenerate matrix, > 2^31 - 1 under Node 12): same error here, before anything reaches the native side.
function checkedLength(n) { if (n > 2147483647) throw new RangeError(`Invalid typed array length: ${n}`); return n; }
const asF32 = (p) => (p instanceof Float32Array ? p : Float32Array.from(p));     // what the native side needs (values are already f32 when typed Float32Array)
// value equality of two point lists as the reference's Float32Array scratch triangles see them (:792-799); NaN equals NaN
function samePoints(a, b) {
    if (a.length !== b.length) return false;
    for (let i = 0; i < a.length; i++) { const x = Math.fround(a[i]), y = Math.fround(b[i]); if (x !== y && (x === x || y === y)) return false; }
    return true;
}
function sameTriangles(a, b) {
    if (a === b) return true;
    if (a === null || b === null || a.length !== b.length) return false;
    for (let i = 0; i < a.length; i++) if (a[i] !== b[i]) return false;
    return true;
}
// `new Int16Array(n)` of the map builders (:820, :848): a negative length throws, NaN makes an empty array
function checkedMapLength(n) { if (n < 0 || n > 2147483647) throw new RangeError(`Invalid typed array length: ${n}`); return n; }

class Homography {
    constructor(transform = 'auto', width = null, height = null, options = {}) {
        if (width !== null) width = Math.round(width);          // :80-81
        if (height !== null) height = Math.round(height);
        this._width = width; this._height = height;
        this._objectiveWidth = null; this._objectiveHeight = null;
        this._xOutputOffset = undefined; this._yOutputOffset = undefined;
        this._srcPoints = null; this._dstPoints = null;
        this.firstTransformSelected = transform.toLowerCase();
        this.transform = transform.toLowerCase();
        this._image = null;
        this._maxSrcX = null; this._maxSrcY = null; this._minSrcX = null; this._minSrcY = null;
        this._srcPointsAreNormalized = true; this._dstPointsAreNormalized = true;
        // The two caches of the reference's piecewise path, mirrored as VALUE SNAPSHOTS of what they were computed from (the arrays
        // themselves live on the GPU and are rebuilt there):
        //   _map  <-> _trianglesCorrespondencesMatrix (:115): null | {kind: 'forward' | 'inverse', pts, tris, width, height, yOff} = the
        //             point set, triangles and geometry fillTriangle rasterised (:817-832 source points over the source bbox, :845-861
        //             destiny points over the output window).  Its null-ness steers re-triangulation (:252, :756); a forward warp reads
        //             WHATEVER it holds (:957);
        //   _pm   <-> _piecewiseMatrices (:119): null | {src, dst, tris} = the point sets and triangles the per-triangle matrices
        //             were solved from (:785-804); both warps use those matrices as they stand (:961, :1036-1038).
        this._map = null;
        this._pm = null;
        this._triangles = null;
        this._transformMatrix = null;
        // Opt-out of the stale-state quirks (header): forward warps always read the forward map of the current mesh, every warp the
        // matrices of the current point sets.
        this.repairStaleMap = options.repairStaleMap === true;
        this._lastPath = null;
        this._native = addon();                                 // throws if the addon is missing: there is no JS fallback
        this._device = options.device === undefined ? 0 : options.device;
        // Opt-in (not in the reference): warp() writes every frame of the inverse paths into one internal buffer and returns
        // a view of it, instead of allocating a fresh 4*w*h-byte array per call -- for frame loops that consume a frame
        // before asking for the next one.  Default false: a fresh array per call, like the reference (:991, :1040).
        this.reuseOutput = options.reuseOutput === true;
        this._outBuffer = null;
        // Opt-in (not in the reference), the same idea for warpBatch(): the frames of a batch are views of ONE page-locked buffer owned by
        // this instance and the NEXT warpBatch() on it overwrites them.  Default false: every frame owns its buffer, like the frames the
        // reference's loop returns -- `out.push(...h.warpBatch(chunk))` keeps working.  (Per call: warpBatch(sets, {reuseBatchOutput: true}).)
        this.reuseBatchOutput = options.reuseBatchOutput === true;
        // Opt-in (not in the reference): the caller promises not to mutate `image.data` between warps, so the source is
        // uploaded once per setImage() / context instead of on every warp().  Default false: the reference aliases the
        // caller's buffer and re-reads it on every warp (:298), so a mutated buffer must show up.
        this.staticImage = options.staticImage === true;
        this._uploadedImage = null;
        this._ctxHandle = null;                                 // GPU context, created at the first warp
        this._multiHandle = null; this._multiKey = null; this._multiImage = null;   // hg_multi over a device list (warpBatch({devices}))
    }

    /** Which map the shared field would hold: null | 'forward' | 'inverse'. */
    get _mapState() { return this._map === null ? null : this._map.kind; }

    /** GPU context of this instance (one hg_ctx per Homography).  Throws a string without a usable gfx950 device. */
    get _ctx() {
        if (this._ctxHandle === null) this._ctxHandle = this._native.create(this._device);
        return this._ctxHandle;
    }

    /** Frees the GPU context (optional; it is also released when the object is garbage-collected). */
    close() {
        if (this._ctxHandle) { this._native.destroy(this._ctxHandle); this._ctxHandle = null; }
        if (this._multiHandle) { this._native.multiDestroy(this._multiHandle); this._multiHandle = null; this._multiKey = null; }
    }

    // ------------------------------------------------------------------------------------------------ public setters
    setReferencePoints(srcPoints, dstPoints, image = null, width = null, height = null, srcPointsAreNormalized = null, dstPointsAreNormalized = null) {   // :173-182
        if (typeof (srcPoints) === 'undefined' || typeof (dstPoints) === 'undefined') {
            throw ("Source and Destiny points must be defined when calling setReferencePoints().");
        }
        this._dstPoints = null;
        this.setSourcePoints(srcPoints, image, width, height, srcPointsAreNormalized);
        this.setDestinyPoints(dstPoints, dstPointsAreNormalized);
    }

    setSourcePoints(points, image = null, width = null, height = null, pointsAreNormalized = null) {    // :218-265
        points = toF32(points);
        this._srcPoints = points;
        this._srcPointsAreNormalized = pointsAreNormalized === null ? !anyAbove(points, NORMALIZED_MAX) : pointsAreNormalized;
        this._transformMatrix = null;
        this.transform = selectTransform(this.firstTransformSelected, points);
        this._objectiveWidth = null; this._objectiveHeight = null;
        if (image !== null) this.setImage(image, width, height);
        else if (width !== null || height !== null) this._setSourceSize(width, height);
        if (this._width !== null && this._height !== null && this._srcPointsAreNormalized) {
            scalePoints(this._srcPoints, this._width, this._height);
            this._srcPointsAreNormalized = false;
        }
        if (this._dstPoints !== null && this.transform !== 'piecewiseaffine') {
            this._transformMatrix = this._solve(this._srcPoints, this._dstPoints);
        }
        if (this.transform === 'piecewiseaffine' && this._map === null) {
            this._triangles = null;                             // :254 (_initialTriangles is never set by the reference)
            this._pm = null;                                    // :255 (ONLY here: with a map in place the old matrices survive new source points)
            if (!this._srcPointsAreNormalized || (this._width > 0 && this._height > 0)) this._refreshPiecewise();
            else if (this._triangles === null) this._triangles = Homography.triangulate(this._srcPoints);
        }
    }

    setImage(image, width = null, height = null) {                                                      // :290-316
        if (!image || !ArrayBuffer.isView(image.data)) {
            throw ("hgwarp: setImage() needs an ImageData-shaped {data: Uint8ClampedArray, width, height}; HTMLImageElement inputs need a browser DOM");
        }
        this._image = image.data;                               // aliased, not copied (:298): re-read (re-uploaded) on every warp()
        this._uploadedImage = null;                             // (staticImage: an explicit setImage always uploads)
        this._setSourceSize(image.width, image.height);
        if (this._srcPoints !== null && this.transform === 'piecewiseaffine') this._refreshPiecewise();
        if (this._dstPoints !== null && (this._objectiveWidth <= 0 || this._objectiveHeight <= 0)) this._deriveOutputWindow();
    }

    setDestinyPoints(points, pointsAreNormalized = null) {                                              // :337-380
        points = toF32(points);
        if (this._srcPoints !== null && points.length !== this._srcPoints.length) {
            throw (`It must be the same amount of destiny points (${points.length / 2}) than source points (${this._srcPoints.length / 2})`);
        }
        this._dstPoints = points;
        this._dstPointsAreNormalized = pointsAreNormalized === null ? !anyAbove(points, NORMALIZED_MAX) : pointsAreNormalized;
        if (this.transform !== 'piecewiseaffine') {
            if (this._dstPointsAreNormalized && this._width > 0 && this._height > 0 && this.transform === 'projective') {
                scalePoints(this._dstPoints, this._width, this._height);
                this._dstPointsAreNormalized = false;
            }
            this._alignRanges();
            this._transformMatrix = this._solve(this._srcPoints, this._dstPoints);
        } else {
            this._pm = null;                                    // :361
        }
        if (this._image !== null || (this.transform === 'piecewiseaffine' && this._width > 0 && this._height > 0)) this._deriveOutputWindow();
        if (this.transform === 'piecewiseaffine' && this._width > 0 && this._height > 0) {
            if (this._dstPointsAreNormalized) { scalePoints(this._dstPoints, this._width, this._height); this._dstPointsAreNormalized = false; }
            this._refreshPiecewise();
        }
    }

    setTriangles(triangles) {                                                                           // :517-524
        this._triangles = triangles;
        if ((!this._srcPointsAreNormalized || (this._width > 0 && this._height > 0)) && this._srcPoints !== null) this._refreshPiecewise();
    }

    // ------------------------------------------------------------------------------------------------ warp
    warp(image = null, asHTMLPromise = false, applyAlwaysInverse = false) {                             // :408-446
        if (image !== null) this.setImage(image);
        else if (this._image === null) {
            throw ("warp() must receive an image if it was not setted before through `setImage(img)` or  `setSourcePoints(points, img)`");
        }
        if (asHTMLPromise) throw ("hgwarp: asHTMLPromise needs a browser DOM; use the returned ImageData-shaped object");
        let out;
        switch (this.transform) {
            case 'piecewiseaffine':
                out = (applyAlwaysInverse || (this._objectiveWidth > this._width || this._objectiveHeight > this._height ||
                       this._objectiveWidth * 1.2 < this._width || this._objectiveHeight * 1.2 < this._height))
                    ? this._inversePiecewise() : this._forwardPiecewise();
                break;
            case 'affine':
                out = (applyAlwaysInverse || (this._objectiveWidth !== this._width || this._objectiveHeight !== this._height))
                    ? this._inverseGeometric() : this._forwardGeometric();
                break;
            case 'projective':
                out = this._inverseGeometric();
                break;
        }
        const area = this._objectiveWidth * this._objectiveHeight;
        if (area >= 1 && !isNaN(area)) return makeImageData(out, this._objectiveWidth, this._objectiveHeight);
        return makeImageData(new Uint8ClampedArray(4), 1, 1);                                           // :440
    }

    /**
     * The benchmarked caller loop `for (f) { setDestinyPoints(dst[f]); warp(); }` (test/benchmark.js:107-110) as GPU batches.
     * dstPointSets: array of point sets.  Returns an array of ImageData-shaped frames, frame f identical to what the loop returns
     * for it -- INCLUDING warp()'s choice between the inverse and the forward (scatter-semantics) loop, made per frame on that
     * frame's output window exactly as warp() makes it (:421-422 piecewise, :426-427 affine, :431 projective): the frames that
     * dispatch inverse go through one inverse batch, the ones that dispatch forward through one forward batch.
     * options: {inverse: true} = the loop with warp(null, false, true) (applyAlwaysInverse: every frame through the inverse loop);
     *          {images: [...]} = the loop warp(image_f) (frame f reads images[f % images.length]); {devices: [...]} = the inverse
     *          frames spread over several GPUs of this node (forward frames run on this instance's own device);
     *          {pointsAreNormalized: bool} = the second argument of every setDestinyPoints (default: the reference's auto-detect).
     *          {reuseBatchOutput: true} (or the constructor option of that name) = the frames are views of one buffer of this instance that
     *          the next warpBatch() overwrites (see below); {ownFrames: true} overrides it for one call.
     * Every setDestinyPoints(dst[f]) runs on the host exactly as in the loop (normalisation auto-detect, in-place scaling of typed
     * arrays, window derivation), so the instance ends in the state the loop would leave it in.
     * LIFE TIME OF THE FRAMES: by default every frame owns its buffer (a pooled page-locked one while the pool has room, else a V8 array),
     * exactly like the frames the reference's loop returns -- a caller may accumulate them across calls.  With `reuseBatchOutput` (opt-in,
     * like `reuseOutput` for warp()) the frames of a batch are views of ONE page-locked buffer owned by this instance, reused by the next
     * warpBatch() on it: consume (or copy) a batch before asking for the next one.  The returned array then has a `release()` method that
     * gives the buffer back at once (the frames become empty), and nothing depends on when V8 collects garbage: no 34-MB allocation per 4K
     * frame, no fall-back to plain arrays in a loop that never yields (plain `node`, 4K, batches of 8: 0.73 instead of 2.05 ms per frame).
     * The slab counts against Homography.setPinnedLimit(); when it does not fit (or the limit is 0) the batch silently gets own frames.
     * With {images} the pass is pipelined: the upload of source f + 1 overlaps the download of frame f.
     */
    warpBatch(dstPointSets, options = {}) {
        if (this.transform === 'affine' || this.transform === 'projective') return this._warpBatchGeometric(dstPointSets, options);
        if (this.transform !== 'piecewiseaffine') throw ("hgwarp: warpBatch() needs a transform (set the source points first)");
        const F = dstPointSets.length, n = this._srcPoints.length;
        const all = new Float32Array(F * n), geoms = new Int32Array(F * 4), forward = new Array(F).fill(false), blank = new Array(F).fill(false);
        const stale = [];                                       // forward frames that read something else than the forward map of the current mesh (:957)
        for (let f = 0; f < F; f++) {
            this.setDestinyPoints(dstPointSets[f], options.pointsAreNormalized === undefined ? null : options.pointsAreNormalized);   // :337-380, as the loop does
            if (this._image === null && !options.images) throw ("warp() must receive an image if it was not setted before through `setImage(img)` or  `setSourcePoints(points, img)`");   // the loop's first warp() (:411-413)
            all.set(asF32(this._dstPoints), f * n);
            const [xo, yo, ow, oh] = this._window();
            forward[f] = !(options.inverse === true || ow > this._width || oh > this._height || ow * 1.2 < this._width || oh * 1.2 < this._height);   // :421-422
            blank[f] = !(ow * oh >= 1);                                                                 // :440: a 1 x 1 blank frame
            // the shared map field, carried from frame to frame as the loop would: an inverse frame leaves its own map there (:847-857),
            // a forward frame reads what it finds
            if (!forward[f]) {
                checkedMapLength(ow * oh);
                this._map = { kind: 'inverse', pts: all.subarray(f * n, (f + 1) * n), tris: this._triangles, width: ow, height: oh, yOff: yo };
            } else if (this.repairStaleMap) {
                if (!this._mapIsCurrentForward()) this._snapshotForwardMap();
            } else if (!(this._mapIsCurrentForward() && this._matricesAreCurrent())) {
                if (this._map === null) throw new TypeError("Cannot read property '0' of null");
                stale.push({ f, map: this._map, mats: this._snapshotMatrices(), blank: blank[f] });     // (a blank one too: its loop still reads the held map, :957-961)
                blank[f] = blank[f] || null;                                                            // (null: neither blank nor batched)
            }
            if (blank[f] === true) continue;
            checkedLength(ow * oh * 4);
            geoms.set([xo, yo, ow, oh], f * 4);
        }
        const frames = new Array(F).fill(null);
        const pick = (want) => { const ids = []; for (let f = 0; f < F; f++) if (blank[f] === false && forward[f] === want) ids.push(f); return ids; };
        const subset = (ids) => {                                                                       // points / windows / sources of a subset of the frames
            if (ids.length === F) return { pts: all, g: geoms, images: options.images };
            const pts = new Float32Array(ids.length * n), g = new Int32Array(ids.length * 4);
            ids.forEach((f, k) => { pts.set(all.subarray(f * n, (f + 1) * n), k * n); g.set(geoms.subarray(4 * f, 4 * f + 4), 4 * k); });
            const images = options.images ? ids.map((f) => options.images[f % options.images.length]) : options.images;
            return { pts, g, images };
        };
        const own = options.ownFrames === true || !(options.reuseBatchOutput === undefined ? this.reuseBatchOutput : options.reuseBatchOutput === true);
        const room = (g) => { if (!own) return; let largest = 0; for (let k = 0; k < g.length / 4; k++) largest = Math.max(largest, g[4 * k + 2] * g[4 * k + 3] * 4); makeRoomFor(this._native, largest, g.length / 4); };
        const inv = pick(false), fwd = pick(true);
        if (inv.length) {
            const { pts, g, images } = subset(inv);
            room(g);
            let datas;
            if (options.devices !== undefined && options.devices !== null) {
                // several GPUs of this node: device i of G warps a contiguous block of the frames (hg_multi_*: no collective on the
                // data path; the shared source is fanned out once over xGMI peer copies -- or, with {images}, every device uploads the
                // sources of its own block and nothing is exchanged at all)
                const multi = this._multiFor(options.devices);
                const tris = this._triangles instanceof Uint32Array ? this._triangles : Uint32Array.from(this._triangles);
                this._native.multiSetMesh(multi, asF32(this._srcPoints), tris, this._minSrcX, this._minSrcY);
                if (images) {
                    datas = this._native.multiWarpBatch(multi, pts, g, this._checkedSources(images), this._width, this._height);
                    this._multiImage = null;                                                             // (the devices now hold the per-frame sources)
                } else {
                    if (!(this.staticImage && this._multiImage === this._image)) {
                        this._native.multiSetImage(multi, this._image, this._width, this._height);
                        this._multiImage = this._image;
                    }
                    datas = this._native.multiWarpBatch(multi, pts, g);
                }
            } else {
                this._uploadMesh();
                datas = this._batchSources(images, (...src) => this._native.warpInversePiecewiseBatch(this._ctx, pts, g, own, ...src));
            }
            inv.forEach((f, k) => { frames[f] = makeImageData(datas[k], g[4 * k + 2], g[4 * k + 3]); });
        }
        if (fwd.length) {                                                                               // _piecewiseAffineWarp :948-972 for the frames warp() sends there
            const { pts, g, images } = subset(fwd);
            room(g);
            this._uploadMesh();
            const datas = this._batchSources(images, (...src) => this._native.warpForwardPiecewiseBatch(this._ctx, pts, this._maxSrcX, this._maxSrcY, g, own, ...src));
            fwd.forEach((f, k) => { frames[f] = makeImageData(datas[k], g[4 * k + 2], g[4 * k + 3]); });
        }
        for (const { f, map, mats, blank: isBlank } of stale) {                                         // forward frames over a stale map: one by one, as they stand
            if (isBlank) { if (options.images) this._uploadSources([options.images[f % options.images.length]]); else this._uploadImage(); this._forwardOverHeldMap(mats, map, 0, 0, 0, 0); continue; }
            const [xo, yo, ow, oh] = geoms.subarray(4 * f, 4 * f + 4);
            makeRoomFor(this._native, ow * oh * 4, 1);
            if (options.images) this._uploadSources([options.images[f % options.images.length]]); else this._uploadImage();
            frames[f] = makeImageData(this._forwardOverHeldMap(mats, map, xo, yo, ow, oh), ow, oh);
        }
        for (let f = 0; f < F; f++) if (blank[f] === true) frames[f] = makeImageData(new Uint8ClampedArray(4), 1, 1);
        if (F > 0) this._lastPath = forward[F - 1] ? '_piecewiseAffineWarp' : '_inversePiecewiseAffineWarp';   // what the last warp() of the loop leaves behind
        return this._releasable(frames);
    }

    /**
     * warpBatch() for affine / projective: every `setDestinyPoints(dst_f)` of the loop runs on the host exactly as in the loop
     * (normalisation auto-detect, in-place range alignment, forward matrix, output window).  Frames that warp() sends down the inverse
     * loop (projective always :431; affine when the output size differs from the source's :426, or with {inverse: true}): the inverse
     * matrices the reference re-solves inside every warp (:994) are solved on the GPU, one lane per frame, all frames in one launch.
     * Affine frames of the source's size go through the forward loop _geometricWarp (:427) with their forward matrices, as one batch.
     */
    _warpBatchGeometric(dstPointSets, options = {}) {
        const F = dstPointSets.length, per = this.transform === 'affine' ? 6 : 8;
        const from = new Float32Array(F * per), to = new Float32Array(F * per), geoms = new Int32Array(F * 4), mats = new Float64Array(F * 8);
        const blank = new Array(F).fill(false), forward = new Array(F).fill(false);
        for (let f = 0; f < F; f++) {
            this.setDestinyPoints(dstPointSets[f], options.pointsAreNormalized === undefined ? null : options.pointsAreNormalized);
            if (this._image === null && !options.images) throw ("warp() must receive an image if it was not setted before through `setImage(img)` or  `setSourcePoints(points, img)`");   // the loop's first warp() (:411-413)
            const [xo, yo, ow, oh] = this._window();
            forward[f] = this.transform === 'affine' && options.inverse !== true && ow === this._width && oh === this._height;      // :426-427, :431
            if (forward[f]) mats.set(Array.from(this._transformMatrix), f * 8);                          // _geometricWarp uses _transformMatrix as it stands (:915)
            else {
                this._alignRanges();                                                                     // :993
                from.set(asF32(this._dstPoints).subarray(0, per), f * per);                              // inverse: dst -> src (:994)
                to.set(asF32(this._srcPoints).subarray(0, per), f * per);
            }
            if (!(ow * oh >= 1)) { blank[f] = true; geoms.set([0, 0, 0, 0], f * 4); continue; }
            checkedLength(ow * oh * 4);
            geoms.set([xo, yo, ow, oh], f * 4);
        }
        const kind = this.transform === 'affine' ? AFFINE : PROJECTIVE;
        const frames = new Array(F).fill(null);
        const pick = (want) => { const ids = []; for (let f = 0; f < F; f++) if (!blank[f] && forward[f] === want) ids.push(f); return ids; };
        const own = options.ownFrames === true || !(options.reuseBatchOutput === undefined ? this.reuseBatchOutput : options.reuseBatchOutput === true);
        const room = (g) => { if (!own) return; let largest = 0; for (let k = 0; k < g.length / 4; k++) largest = Math.max(largest, g[4 * k + 2] * g[4 * k + 3] * 4); makeRoomFor(this._native, largest, g.length / 4); };
        const sub = (arr, ids, w) => { if (ids.length === F) return arr; const o = new arr.constructor(ids.length * w); ids.forEach((f, k) => o.set(arr.subarray(f * w, (f + 1) * w), k * w)); return o; };
        const subImages = (ids) => (options.images && ids.length !== F ? ids.map((f) => options.images[f % options.images.length]) : options.images);
        const inv = pick(false), fwd = pick(true);
        if (inv.length) {
            const g = sub(geoms, inv, 4), fr = sub(from, inv, per), tt = sub(to, inv, per), images = subImages(inv);
            room(g);
            let datas;
            if (options.devices !== undefined && options.devices !== null) {                          // frames spread over several GPUs
                const multi = this._multiFor(options.devices);
                if (images) {
                    datas = this._native.multiWarpGeometricBatch(multi, kind, fr, tt, g, this._checkedSources(images), this._width, this._height);
                    this._multiImage = null;
                } else {
                    if (!(this.staticImage && this._multiImage === this._image)) {
                        this._native.multiSetImage(multi, this._image, this._width, this._height);
                        this._multiImage = this._image;
                    }
                    datas = this._native.multiWarpGeometricBatch(multi, kind, fr, tt, g);
                }
            } else {
                datas = this._batchSources(images, (...src) => this._native.warpInverseGeometricBatch(this._ctx, kind, fr, tt, g, own, ...src));
            }
            inv.forEach((f, k) => { frames[f] = makeImageData(datas[k], g[4 * k + 2], g[4 * k + 3]); });
        }
        if (fwd.length) {                                                                               // _geometricWarp :911-932
            const g = sub(geoms, fwd, 4), m = sub(mats, fwd, 8);
            room(g);
            const datas = this._batchSources(subImages(fwd), (...src) => this._native.warpForwardGeometricBatch(this._ctx, kind, m, g, own, ...src));
            fwd.forEach((f, k) => { frames[f] = makeImageData(datas[k], g[4 * k + 2], g[4 * k + 3]); });
        }
        for (let f = 0; f < F; f++) if (blank[f]) frames[f] = makeImageData(new Uint8ClampedArray(4), 1, 1);
        if (F > 0) this._lastPath = forward[F - 1] ? '_geometricWarp' : '_inverseGeometricWarp';
        return this._releasable(frames);
    }

    /** hg_multi handle for a device list (kept while the list stays the same). */
    _multiFor(devices) {
        const ids = Int32Array.from(devices);
        if (ids.length === 0) throw ("hgwarp: warpBatch({devices}) needs at least one device id");
        const key = ids.join(',');
        if (this._multiKey !== key) {
            if (this._multiHandle) this._native.multiDestroy(this._multiHandle);
            this._multiHandle = this._native.multiCreate(ids);
            this._multiKey = key; this._multiImage = null;
        }
        return this._multiHandle;
    }

    getTransformationMatrixAsCSS(srcPoints = null, dstPoints = null, width = null, height = null) {     // :548-587
        if (width !== null || height !== null) this._setSourceSize(width, height);
        if (srcPoints !== null) this.setSourcePoints(srcPoints, null, width, height);
        if (dstPoints !== null) this.setDestinyPoints(dstPoints);
        if (this._srcPoints === null) throw ("Impossible to calculate a transform when srcPoints are not set");
        else if (this._dstPoints === null) throw ("Impossible to calculate a transform when dstPoints are not set");
        else if (this._transformMatrix === null) throw ("Transform matrix can not be calculated");
        const m = this._transformMatrix, fx = (v) => v.toFixed(CSS_DECIMALS);
        if (this.transform === 'affine') return `matrix(${Array.from(m, fx).join(', ')})`;
        if (this.transform === 'projective') {
            const cells = [];
            let i = 0;
            for (let row = 0; row < 4; row++) for (let col = 0; col < 4; col++) {
                if ((row === 2 && col === 2) || (row === 3 && col === 3)) cells.push('1');
                else if (row === 2 || col === 2) cells.push('0');
                else cells.push(fx(m[((i++) * 3) % 8]));
            }
            return `matrix3d(${cells.join(', ')})`;
        }
        throw (`Only "affine" or "projective" transforms can be applied on the CSS transform property, but ${this.transform} selected`);
    }

    HTMLImageElementFromImageData() { throw ("hgwarp: HTMLImageElementFromImageData() needs a browser DOM"); }
    transformHTMLElement() { throw ("hgwarp: transformHTMLElement() needs a browser DOM"); }

    // ------------------------------------------------------------------------------------------------ parity taps (debug)
    /** Int16Array the reference's _buildInverseTrianglesCorrespondencesMatrix would hold for the last inverse piecewise warp. */
    triangleMap(fused = true) { return this._native.getTriMap(this._ctx, !!fused); }
    /** {forward, inverse}: per-triangle Float32Array(6 T) of the last piecewise warp. */
    piecewiseMatrices() { return this._native.getMatrices(this._ctx, this._triangles.length / 3); }

    // ------------------------------------------------------------------------------------------------ private
    _solve(from, to) {                                                                                   // :1237-1250
        // a point set shorter than the transform needs (an 'auto' instance between a 4-point setSourcePoints and the matching
        // setDestinyPoints): the reference reads `undefined` past the end, i.e. NaN in its arithmetic
        const need = this.transform === 'affine' ? 6 : 8;
        const padded = (p) => { p = asF32(p); if (p.length >= need) return p; const q = new Float32Array(need).fill(NaN); q.set(p); return q; };
        const a = padded(from), b = padded(to);
        if (this.transform === 'affine') return this._native.solveAffine(a, b);                          // Float32Array(6)
        if (this.transform === 'projective') return Array.from(this._native.solveProjective(a, b));      // plain Array(8) of doubles, as numeric.js returns
        throw (`${this.transform} transform does not exist`);
    }

    _setSourceSize(width, height) {                                                                      // :637-679
        const lastW = this._width, lastH = this._height;
        this._width = width; this._height = height;
        if (lastW === width && lastH === height) return;
        this._width = Math.round(width); this._height = Math.round(height);
        this._map = null;                                                                                // :647
        if (this.transform === 'projective') {
            if (this._srcPoints !== null && this._srcPointsAreNormalized) { scalePoints(this._srcPoints, this._width, this._height); this._srcPointsAreNormalized = false; }
            if (this._dstPoints !== null && this._dstPointsAreNormalized) { scalePoints(this._dstPoints, this._width, this._height); this._dstPointsAreNormalized = false; }
            if (this._dstPoints !== null && this._srcPoints !== null) {
                this._transformMatrix = this._solve(this._srcPoints, this._dstPoints);
                this._deriveOutputWindow();
            }
        }
        if (this._srcPoints !== null && this.transform === 'piecewiseaffine') this._refreshPiecewise();
    }

    _deriveOutputWindow() {                                                                              // :693-725
        if (this.transform === 'affine' || this.transform === 'projective') {
            if (this._transformMatrix === null) {
                if (this._srcPointsAreNormalized !== this._dstPointsAreNormalized) this._alignRanges();
                this._transformMatrix = this._solve(this._srcPoints, this._dstPoints);
            }
            const lim = this._native.transformLimits(this.transform === 'affine' ? AFFINE : PROJECTIVE, Float64Array.from(this._transformMatrix), this._width, this._height);
            [this._xOutputOffset, this._yOutputOffset, this._objectiveWidth, this._objectiveHeight] = lim;
        } else if (!this._dstPointsAreNormalized) {
            const mm = this._native.minmaxXY(asF32(this._dstPoints));                                    // rounded min/max, then subtract (:707-710)
            this._xOutputOffset = mm[0]; this._yOutputOffset = mm[1];
            this._objectiveWidth = mm[2] - mm[0]; this._objectiveHeight = mm[3] - mm[1];
        } else if (this._width > 0 && this._height > 0) {
            let minX = Infinity, minY = Infinity, maxX = -Infinity, maxY = -Infinity;                    // unrounded (:714-718)
            const p = this._dstPoints;
            for (let i = 0; i < p.length; i++) {
                if ((i % 2) === 0) { if (p[i] > maxX) maxX = p[i]; if (p[i] < minX) minX = p[i]; }
                else { if (p[i] > maxY) maxY = p[i]; if (p[i] < minY) minY = p[i]; }
            }
            this._xOutputOffset = Math.round(minX); this._yOutputOffset = Math.round(minY);
            this._objectiveWidth = Math.round((maxX - minX) * this._width);
            this._objectiveHeight = Math.round((maxY - minY) * this._height);
        } else {
            throw ("Trying to calculate a the output width and height of a Piecewise Affine transform but source width and height are not set");
        }
    }

    _refreshPiecewise() {                                                                                // :738-774
        if (this._srcPoints === null) throw ("Trying to set the Piecewise Affine Transform parameters before setting the Source Points.");
        if (this._triangles === null) this._triangles = Homography.triangulate(this._srcPoints);
        if (this._srcPointsAreNormalized) {
            if (this._width > 0 && this._height > 0) { scalePoints(this._srcPoints, this._width, this._height); this._srcPointsAreNormalized = false; }
            else throw ("Trying to set the Piecewise Affine Transform parameters without knowing the source points ranges");
        }
        if (!this._srcPointsAreNormalized && (this._triangles === null || this._map === null)) {
            const mm = this._native.minmaxXY(asF32(this._srcPoints));                                    // :758
            [this._minSrcX, this._minSrcY, this._maxSrcX, this._maxSrcY] = mm;
            this._snapshotForwardMap();                                                                  // :759
        }
        if (this._dstPoints !== null && this._pm === null && this._triangles !== null) {
            if (this._dstPointsAreNormalized) { scalePoints(this._dstPoints, this._width, this._height); this._dstPointsAreNormalized = false; }
            if (this._srcPointsAreNormalized !== this._dstPointsAreNormalized) this._alignRanges();     // :787-789
            // :769: matrices of THESE point sets (the per-triangle solves run on the GPU with the warp)
            this._pm = { src: Float32Array.from(this._srcPoints), dst: Float32Array.from(this._dstPoints), tris: this._triangles };
        }
    }

    _alignRanges() {                                                                                     // :876-896
        if (this._dstPointsAreNormalized === this._srcPointsAreNormalized) return;
        if (this._dstPointsAreNormalized && this._width > 0 && this._height > 0) {
            unscalePoints(this._srcPoints, this._width, this._height);
            this._srcPointsAreNormalized = true;
        } else if (this._srcPointsAreNormalized && this._width > 0 && this._height > 0) {
            scalePoints(this._srcPoints, this._width, this._height);
            this._srcPointsAreNormalized = false;
        } else {
            throw ("Impossible to put source and destiny points in the same range. Possible solutions: \n" +
                   "1. Give a source width/height when calling setSrcPoints.\n" +
                   "2. Set the input image before.\n" +
                   "3. Give Source and Destiny points in the same range (both normalized or both in image dimensions)");
        }
    }

    _uploadImage() {
        // The reference re-reads the caller's buffer on every warp (:298), so a mutated buffer must show up: upload every time
        // (unless the caller opted into `staticImage`).
        const ctx = this._ctx;
        if (this.staticImage && this._uploadedImage === this._image && this._uploadedCtx === ctx) return;
        this._native.setImage(ctx, this._image, this._width, this._height);
        this._uploadedImage = this._image; this._uploadedCtx = ctx;
    }

    /** Source(s) of a batch: the instance's image, or `images` (one ImageData-shaped source per frame, all of the instance's size:
     *  the loop `warp(image_f)`; frame f reads images[f % images.length]). */
    _checkedSources(images) {
        if (!Array.isArray(images) || images.length === 0) throw ("hgwarp: warpBatch({images}) needs a non-empty array of ImageData-shaped sources");
        for (const im of images) {
            if (!im || !ArrayBuffer.isView(im.data) || im.width !== this._width || im.height !== this._height)
                throw ("hgwarp: every image of warpBatch({images}) must be ImageData-shaped and of the size the instance was set up with");
        }
        return images.map((im) => im.data);
    }

    _uploadSources(images) {
        if (images === undefined || images === null) return this._uploadImage();
        this._native.setImages(this._ctx, this._checkedSources(images), this._width, this._height);
        this._uploadedImage = null;                              // the next single-image warp uploads again
    }

    /** Runs a native batch with the instance's image (uploaded first) or with per-frame sources handed to the native side, which
     *  pipelines their uploads with the downloads of the frames (frame f reads images[f % images.length]). */
    _batchSources(images, run) {
        if (images === undefined || images === null) { this._uploadImage(); return run(); }
        const datas = run(this._checkedSources(images), this._width, this._height);
        this._uploadedImage = null;                              // the next single-image warp uploads again
        return datas;
    }

    /** The frames of a batch + `release()`: the page-locked buffer behind them goes back at once (they become empty). */
    _releasable(frames) {
        Object.defineProperty(frames, 'release', { value: () => { if (this._ctxHandle) this._native.releaseBatch(this._ctxHandle); }, enumerable: false });
        return frames;
    }

    _uploadMesh() {
        const tris = this._triangles instanceof Uint32Array ? this._triangles : Uint32Array.from(this._triangles);
        this._native.piecewiseSetMesh(this._ctx, asF32(this._srcPoints), tris, this._minSrcX, this._minSrcY);
    }

    _window() { return [this._xOutputOffset, this._yOutputOffset, this._objectiveWidth, this._objectiveHeight]; }

    /** The caller-owned output buffer handed to the addon when `reuseOutput` is on (grown as needed), else undefined. */
    _reusable(bytes) {
        if (!this.reuseOutput) return undefined;
        if (this._outBuffer === null || this._outBuffer.length < bytes) this._outBuffer = new Uint8ClampedArray(bytes);
        return this._outBuffer;
    }

    _inverseGeometric() {                                                                                // :987-1013
        this._lastPath = '_inverseGeometricWarp';
        this._alignRanges();
        const inv = this._solve(this._dstPoints, this._srcPoints);                                       // re-solved from the swapped sets (:994)
        this._uploadImage();
        const [xo, yo, ow, oh] = this._window();
        if (!(ow * oh >= 1)) return new Uint8ClampedArray(0);
        checkedLength(ow * oh * 4);
        if (!this.reuseOutput) makeRoomFor(this._native, ow * oh * 4, 1);
        return this._native.warpInverseGeometric(this._ctx, this.transform === 'affine' ? AFFINE : PROJECTIVE, Float64Array.from(inv), xo, yo, ow, oh,
                                                 this._reusable(ow * oh * 4));
    }

    /** Do the cached per-triangle matrices (:769) belong to the current point sets and triangles?  (Always, unless source points or
     *  triangles were replaced after the last setDestinyPoints: :252-255, :519.) */
    _matricesAreCurrent() {
        const pm = this._pm;
        return pm !== null && sameTriangles(pm.tris, this._triangles) && samePoints(pm.src, this._srcPoints) && samePoints(pm.dst, this._dstPoints);
    }

    /** Does the shared map field hold the forward map of the current mesh (:817-832 over the current source bbox)? */
    _mapIsCurrentForward() {
        const m = this._map;
        return m !== null && m.kind === 'forward' && m.width === this._maxSrcX - this._minSrcX && m.height === this._maxSrcY - this._minSrcY &&
               m.yOff === this._minSrcY && sameTriangles(m.tris, this._triangles) && samePoints(m.pts, this._srcPoints);
    }

    /** Forward matrices of the snapshot the reference's `_piecewiseMatrices` came from, 6 floats per triangle (host-side solves). */
    _snapshotMatrices() {
        const pm = this._pm;
        if (pm === null) throw new TypeError("Cannot read property 'length' of null");                  // this._piecewiseMatrices.length (:1036) / [inTriangle] (:961)
        const tris = pm.tris instanceof Uint32Array ? pm.tris : Uint32Array.from(pm.tris);
        return this._native.solveAffineTriangles(pm.src, pm.dst, tris);
    }

    _inversePiecewise() {                                                                                // :1029-1058
        this._lastPath = '_inversePiecewiseAffineWarp';
        const [xo, yo, ow, oh] = this._window();
        checkedMapLength(ow * oh);                                                                       // :848
        // :847-857: the shared field now holds the INVERSE map of these destiny points over this window
        this._map = { kind: 'inverse', pts: Float32Array.from(this._dstPoints), tris: this._triangles, width: ow, height: oh, yOff: yo };
        const current = this.repairStaleMap || this._matricesAreCurrent();
        const mats = current ? null : this._snapshotMatrices();
        if (!(ow * oh >= 1)) return new Uint8ClampedArray(0);
        checkedLength(ow * oh * 4);
        this._uploadImage();
        if (!this.reuseOutput) makeRoomFor(this._native, ow * oh * 4, 1);
        if (!current) {
            // matrices of an OLDER point set / triangle list over the map of the current one (setSourcePoints or setTriangles without a
            // setDestinyPoints since): the reference's loop as it stands, through the materialised map
            const tris = this._triangles instanceof Uint32Array ? this._triangles : Uint32Array.from(this._triangles);
            return this._native.warpInversePiecewiseState(this._ctx, mats, asF32(this._dstPoints), tris, this._minSrcX, this._minSrcY, xo, yo, ow, oh);
        }
        this._uploadMesh();
        this._native.piecewisePrepare(this._ctx, asF32(this._dstPoints), xo, yo, ow, oh);
        return this._native.warpInversePiecewise(this._ctx, this._reusable(ow * oh * 4));
    }

    _forwardGeometric() {                                                                                // :911-932
        this._lastPath = '_geometricWarp';
        this._uploadImage();
        const [xo, yo, ow, oh] = this._window();
        if (!(ow * oh >= 1)) return new Uint8ClampedArray(0);
        checkedLength(ow * oh * 4);
        makeRoomFor(this._native, ow * oh * 4, 1);
        return this._native.warpForwardGeometric(this._ctx, this.transform === 'affine' ? AFFINE : PROJECTIVE, Float64Array.from(this._transformMatrix), xo, yo, ow, oh);
    }

    _forwardPiecewise() {                                                                                // :948-972
        this._lastPath = '_piecewiseAffineWarp';
        const [xo, yo, ow, oh] = this._window();
        const usual = this.repairStaleMap || (this._mapIsCurrentForward() && this._matricesAreCurrent());
        const map = this._map, mats = usual ? null : this._snapshotMatrices();
        if (!usual && map === null) throw new TypeError("Cannot read property '0' of null");              // :957 on a null map
        if (this.repairStaleMap && !this._mapIsCurrentForward()) this._snapshotForwardMap();
        this._uploadImage();
        if (!(ow * oh >= 1)) {
            // a blank window (:440) does not stop the reference's loop: it walks the source bbox over the held map and throws where a
            // cell names a matrix that does not exist -- the stale state still gets that check
            if (!usual) this._forwardOverHeldMap(mats, map, 0, 0, 0, 0);
            return new Uint8ClampedArray(0);
        }
        checkedLength(ow * oh * 4);
        makeRoomFor(this._native, ow * oh * 4, 1);
        if (!usual) {
            // :957 reads whatever the shared field holds -- after an inverse warp the stale INVERSE map of that warp's destiny points,
            // laid out objectiveWidth cells per row but indexed (maxSrcX - minSrcX) per row, cells past its end `undefined` -- and :961
            // the matrices as last solved: both handed over as they stand
            return this._forwardOverHeldMap(mats, map, xo, yo, ow, oh);
        }
        this._uploadMesh();
        return this._native.warpForwardPiecewise(this._ctx, asF32(this._dstPoints), this._maxSrcX, this._maxSrcY, xo, yo, ow, oh);
    }

    /** _piecewiseAffineWarp :948-972 over the map snapshot `map` (whatever the shared field held) with the matrices `mats` as last solved. */
    _forwardOverHeldMap(mats, map, xo, yo, ow, oh) {
        const tris = map.tris instanceof Uint32Array ? map.tris : Uint32Array.from(map.tris);
        // (`new Int16Array(w * h)` of a window that did not exist yet -- null, NaN -- is an empty array: every read is `undefined`)
        const held = map.width * map.height >= 1 ? [map.width, map.height, map.yOff] : [0, 0, 0];
        return this._native.warpForwardPiecewiseState(this._ctx, mats, map.pts, tris, held[0], held[1], held[2],
                                                      this._minSrcX, this._minSrcY, this._maxSrcX, this._maxSrcY, xo, yo, ow, oh);
    }

    /** :759 / :817-832: the shared field becomes the forward map of the CURRENT source points over the current source bbox (built on
     *  the GPU when a forward warp needs it). */
    _snapshotForwardMap() {
        const mw = this._maxSrcX - this._minSrcX, mh = this._maxSrcY - this._minSrcY;
        checkedMapLength(mw * mh);
        this._map = { kind: 'forward', pts: Float32Array.from(this._srcPoints), tris: this._triangles, width: mw, height: mh, yOff: this._minSrcY };
    }
}

function selectTransform(transform, points) {                                                            // :1444-1481
    switch (transform) {
        case 'auto':
            if (points.length === 6) return 'affine';
            if (points.length === 8) return 'projective';
            if (points.length > 8) return 'piecewiseaffine';
            throw (`Transforms must contain at least 3 points but only ${points.length / 2} were given`);
        case 'piecewiseaffine':
            if (points.length < 6) throw (`A piecewise (or affine) transform needs to determine least three reference points but only ${points.length / 2} were given`);
            return transform;
        case 'affine':
            if (points.length !== 6) throw (`An affine transform needs to determine exactly three reference points but ${points.length / 2} were given`);
            return transform;
        case 'projective':
            if (points.length !== 8) throw (`A projective transform needs to determine exactly four reference points but ${points.length / 2} were given`);
            return transform;
        default:
            throw (`Transform "${transform}" is unknown`);
    }
}

/** Triangulator used where the reference calls `new Delaunator(points).triangles` (:1216-1218).  Replaceable. */
Homography.triangulate = defaultTriangulate;
/** Optional: hands the pooled page-locked buffer of a frame returned by warp() / warpBatch() back at once (its data becomes
 *  empty).  Without it the buffer returns when the frame is garbage-collected. */
Homography.release = (imageData) => addon().release(imageData && imageData.data ? imageData.data : imageData);
/** An ImageData-shaped, zero-filled SOURCE image whose pixels live in page-locked memory (to be filled by the caller: decoded video frames
 *  ...): uploads out of it are asynchronous DMA at the full PCIe rate -- with warpBatch({images}) the upload of source f + 1 then really
 *  overlaps the download of frame f.  Images in ordinary V8 memory work everywhere too (the runtime stages them: ~20 % slower uploads). */
Homography.pinnedImage = (width, height) => makeImageData(addon().pinnedBuffer(width * height * 4), width, height);
/** Cap of the page-locked frame pool in bytes (default 2 GiB; 0: plain V8 arrays only).  Returns the bytes currently pinned. */
Homography.setPinnedLimit = (bytes) => addon().setPinnedLimit(bytes);
Homography.poolStats = () => addon().poolStats();
/** GPUs visible to the library. */
Homography.deviceCount = () => addon().deviceCount();
Homography.availableTransforms = TRANSFORMS;

export { Homography };
