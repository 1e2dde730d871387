// reference_patch.mjs -- INTEGRATION.md §B as code: the reference's OWN class (Eric-Canas/Homography.js v1.8.0) with its four private
// pixel loops replaced by calls into the N-API addon (lib/hgwarp.node -> libhgwarp.so -> HIP kernels).
//
//     import { Homography as Reference } from 'homography';           // the reference, unmodified
//     import { patchReference } from './homography.js_amd/js/reference_patch.mjs';
//     const Homography = patchReference(Reference, require('./homography.js_amd/lib/hgwarp.node'));
//
// Everything else -- constructor, setters, normalisation auto-detect, output-window logic, warp()'s dispatch, ImageData wrapping -- stays
// the reference's code and runs unchanged.  What a maintainer would paste into Homography.js is exactly the bodies below (with `hg`
// imported at the top of the file and `this._hg = hg.create(0)` in the constructor :78); here they arrive as a subclass, so that the
// unpatched class stays available beside it (tests/js/ref_patched_over_addon.mjs runs both, op by op, and compares the bytes).
//
// file:line = the reference's Homography.js.  The two caches the reference's loops read AS THEY STAND are kept:
//   * `_piecewiseMatrices` (:769) -- still solved by the reference's own `_calculatePiecewiseAffineTransformMatrices` (T tiny solves on
//     the host); the patch only remembers which point sets / triangles they were solved from;
//   * `_trianglesCorrespondencesMatrix` (:819-820 forward map, :847-848 inverse map: ONE field) -- allocated exactly as before (same
//     null-ness, same length, same RangeError for impossible sizes) but no longer rasterised on the host: the patch remembers what it WOULD
//     hold (point set, triangles, width, height, yOffset) and the GPU rasterises it when a loop needs it.
// A loop whose caches belong to the current mesh takes the fast calls (`piecewiseSetMesh` + `piecewisePrepare` + `warpInversePiecewise`,
// `warpForwardPiecewise`); any other state -- stale matrices after setSourcePoints / setTriangles, the inverse map left in the shared
// field when a forward warp follows an inverse one (SURVEY.md Appendix A-Q12) -- goes through the reference-state entry points
// (`warp*PiecewiseState`), which take the matrices and the map's definition as they stand.  Either way: the reference's bytes.
const AFFINE = 0, PROJECTIVE = 1;

const f32 = (p) => (p instanceof Float32Array ? p : Float32Array.from(p));
const u32 = (t) => (t instanceof Uint32Array ? t : Uint32Array.from(t));
function samePoints(a, b) {                                   // as the Float32Array scratch triangles :792-799 see them; NaN equals NaN
    if (a === null || b === null || a.length !== b.length) return false;
    for (let i = 0; i < a.length; i++) { const x = Math.fround(a[i]), y = Math.fround(b[i]); if (x !== y && (x === x || y === y)) return false; }
    return true;
}
function sameTriangles(a, b) {
    if (a === b) return true;
    if (a === null || b === null || a.length !== b.length) return false;
    for (let i = 0; i < a.length; i++) if (a[i] !== b[i]) return false;
    return true;
}
const checkedLength = (n) => { if (n > 2147483647) throw new RangeError(`Invalid typed array length: ${n}`); return n; };     // new Uint8ClampedArray(n) :916 / :952 / :991 / :1040

/**
 * @param {Function} Reference  the reference's Homography class
 * @param {object}   hg         the addon (require('.../lib/hgwarp.node'))
 * @param {object}   [fns]      `calculateTransformMatrix`: the reference's module-private function of that name (:1237-1250), which an
 *                              in-file patch calls directly (:994).  From outside the module it is not reachable: without it the same
 *                              solve runs through the addon's host functions (hg.solveAffine / hg.solveProjective: same bits).
 * @param {object}   [opts]     `device`: GPU index (default 0)
 */
export function patchReference(Reference, hg, fns = {}, opts = {}) {
    const device = opts.device || 0;
    return class Homography extends Reference {
        constructor(...args) {
            super(...args);
            this._hg = hg.create(device);                     // :78  one GPU context per instance
            this._pmFrom = null;                              // what _piecewiseMatrices was solved from
            this._mapFrom = null;                             // what _trianglesCorrespondencesMatrix would hold
        }
        close() { if (this._hg) { hg.destroy(this._hg); this._hg = null; } }

        // ---- the two caches: kept, described, not rasterised on the host
        _calculatePiecewiseAffineTransformMatrices() {        // :785-804 unchanged + a note of its inputs
            const m = super._calculatePiecewiseAffineTransformMatrices();
            this._pmFrom = { src: Float32Array.from(this._srcPoints), dst: Float32Array.from(this._dstPoints), tris: Uint32Array.from(this._triangles) };
            return m;
        }
        _buildTrianglesCorrespondencesMatrix() {              // :817-832 without :822-830
            const w = this._maxSrcX - this._minSrcX, h = this._maxSrcY - this._minSrcY, n = w * h;
            if (this._trianglesCorrespondencesMatrix === null || this._trianglesCorrespondencesMatrix.length !== n) this._trianglesCorrespondencesMatrix = new Int16Array(n);
            this._mapFrom = { kind: 'forward', pts: Float32Array.from(this._srcPoints), tris: Uint32Array.from(this._triangles), width: w, height: h, yOff: this._minSrcY };
            return this._trianglesCorrespondencesMatrix;
        }
        _buildInverseTrianglesCorrespondencesMatrix() {       // :845-861 without :850-858
            const n = this._objectiveWidth * this._objectiveHeight;
            if (this._trianglesCorrespondencesMatrix === null || this._trianglesCorrespondencesMatrix.length !== n) this._trianglesCorrespondencesMatrix = new Int16Array(n);
            this._mapFrom = { kind: 'inverse', pts: Float32Array.from(this._dstPoints), tris: Uint32Array.from(this._triangles), width: this._objectiveWidth,
                              height: this._objectiveHeight, yOff: this._yOutputOffset };
            return this._trianglesCorrespondencesMatrix;
        }
        _matricesAreCurrent() {
            const p = this._pmFrom;
            return this._piecewiseMatrices !== null && p !== null && sameTriangles(p.tris, this._triangles) && samePoints(p.src, this._srcPoints) && samePoints(p.dst, this._dstPoints);
        }
        _mapIsCurrentForward() {
            const m = this._mapFrom;
            return this._trianglesCorrespondencesMatrix !== null && m !== null && m.kind === 'forward' && m.width === this._maxSrcX - this._minSrcX &&
                   m.height === this._maxSrcY - this._minSrcY && m.yOff === this._minSrcY && sameTriangles(m.tris, this._triangles) && samePoints(m.pts, this._srcPoints);
        }
        _matricesAsTheyStand() {                              // 6 floats per triangle: what :961 / :1036-1038 read
            const pm = this._piecewiseMatrices, flat = new Float32Array(6 * pm.length);      // (null: the TypeError of `.length` :1036)
            for (let i = 0; i < pm.length; i++) flat.set(pm[i], 6 * i);
            return flat;
        }

        // ---- the four loops
        _inverseGeometricWarp(image) {                        // :987-1013
            this._putSrcAndDstPointsInSameRange();            // :993 unchanged
            const inv = fns.calculateTransformMatrix ? fns.calculateTransformMatrix(this.transform, this._dstPoints, this._srcPoints)           // :994 unchanged
                                                     : (this.transform === 'affine' ? hg.solveAffine(f32(this._dstPoints), f32(this._srcPoints))
                                                                                    : hg.solveProjective(f32(this._dstPoints), f32(this._srcPoints)));
            const ow = this._objectiveWidth, oh = this._objectiveHeight;
            if (!(ow * oh >= 1)) return new Uint8ClampedArray(0);                 // (warp() :436-441 then returns its 1 x 1 frame)
            checkedLength(ow * oh * 4);
            hg.setImage(this._hg, image, this._width, this._height);
            return hg.warpInverseGeometric(this._hg, this.transform === 'affine' ? AFFINE : PROJECTIVE, Float64Array.from(inv),
                                           this._xOutputOffset, this._yOutputOffset, ow, oh);
        }

        _inversePiecewiseAffineWarp(image) {                  // :1029-1058
            this._buildInverseTrianglesCorrespondencesMatrix();                   // :1033 (allocation + definition; the GPU rasterises)
            const current = this._matricesAreCurrent();
            const mats = current ? null : this._matricesAsTheyStand();            // :1036-1038 read them as they stand
            const xo = this._xOutputOffset, yo = this._yOutputOffset, ow = this._objectiveWidth, oh = this._objectiveHeight;
            if (!(ow * oh >= 1)) return new Uint8ClampedArray(0);
            checkedLength(ow * oh * 4);
            hg.setImage(this._hg, image, this._width, this._height);
            if (!current)
                return hg.warpInversePiecewiseState(this._hg, mats, f32(this._dstPoints), u32(this._triangles), this._minSrcX, this._minSrcY, xo, yo, ow, oh);
            hg.piecewiseSetMesh(this._hg, f32(this._srcPoints), u32(this._triangles), this._minSrcX, this._minSrcY);
            hg.piecewisePrepare(this._hg, f32(this._dstPoints), xo, yo, ow, oh);  // :785-804, :845-861, :1036-1038 on the GPU
            return hg.warpInversePiecewise(this._hg);                             // :1042-1056
        }

        _geometricWarp(image) {                               // :911-932
            const ow = this._objectiveWidth, oh = this._objectiveHeight;
            if (!(ow * oh >= 1)) return new Uint8ClampedArray(0);
            checkedLength(ow * oh * 4);
            hg.setImage(this._hg, image, this._width, this._height);
            return hg.warpForwardGeometric(this._hg, this.transform === 'affine' ? AFFINE : PROJECTIVE, Float64Array.from(this._transformMatrix),
                                           this._xOutputOffset, this._yOutputOffset, ow, oh);
        }

        _piecewiseAffineWarp(image) {                         // :948-972
            const xo = this._xOutputOffset, yo = this._yOutputOffset, ow = this._objectiveWidth, oh = this._objectiveHeight;
            const usual = this._mapIsCurrentForward() && this._matricesAreCurrent();
            const mats = usual ? null : this._matricesAsTheyStand();              // :961
            if (!usual && this._trianglesCorrespondencesMatrix === null) throw new TypeError("Cannot read property '0' of null");      // :957 on a null map
            if (!(ow * oh >= 1)) return new Uint8ClampedArray(0);
            checkedLength(ow * oh * 4);
            hg.setImage(this._hg, image, this._width, this._height);
            if (!usual) {                                     // whatever the shared field holds, indexed as :957 indexes it, and the matrices as last solved
                const m = this._mapFrom, held = m.width * m.height >= 1 ? [m.width, m.height, m.yOff] : [0, 0, 0];
                return hg.warpForwardPiecewiseState(this._hg, mats, m.pts, m.tris, held[0], held[1], held[2],
                                                    this._minSrcX, this._minSrcY, this._maxSrcX, this._maxSrcY, xo, yo, ow, oh);
            }
            hg.piecewiseSetMesh(this._hg, f32(this._srcPoints), u32(this._triangles), this._minSrcX, this._minSrcY);
            return hg.warpForwardPiecewise(this._hg, f32(this._dstPoints), this._maxSrcX, this._maxSrcY, xo, yo, ow, oh);
        }
    };
}
