// Loads the *reference* Homography.js (read-only, from /root/reference) under Node through the three shims of SURVEY.md Appendix B:
// a fake `document`, an `ImageData` class, and a Delaunator stub that returns harness-supplied triangles (globalThis.__TRI__).
// RUNS ONLY IN THE BUILD CONTAINER: nothing of the reference's source text is written into the repository -- the patched copy lives in
// a temp directory for the lifetime of the process.  Users: tests/golden/gen_golden.mjs, tests/js/fuzz_ref_sequences.mjs.
import fs from 'fs';
import os from 'os';
import path from 'path';
import { pathToFileURL } from 'url';

export const REF = process.env.HG_REFERENCE || '/root/reference';
export const referenceAvailable = () => fs.existsSync(path.join(REF, 'Homography.js'));

/** Returns {M, Homography, cleanup}: M = the module namespace (class + the module-private pure functions, exported from the temp copy). */
export async function loadReference() {
    const tmp = fs.mkdtempSync(path.join(os.tmpdir(), 'hgref-'));
    fs.writeFileSync(path.join(tmp, 'package.json'), '{"type":"module"}');
    let src = fs.readFileSync(path.join(REF, 'Homography.js'), 'utf8');
    const importLine = /^\s*import Delaunator from 'https:[^']*';\s*$/m;
    if (!importLine.test(src)) throw new Error('reference import line not found');
    src = src.replace(importLine, "import Delaunator from './delaunator_stub.js';");
    // expose the module-private pure functions for per-function vectors (temp copy only)
    src += '\nexport {fillTriangle, affineMatrixFromTriangles, inverseAffineMatrix, projectiveMatrixFromSquares, ' +
           'calculateTransformMatrix, calculateTransformLimits, minmaxXYofArray, applyAffineTransformToPoint, ' +
           'applyProjectiveTransformToPoint};\n';
    fs.writeFileSync(path.join(tmp, 'Homography.js'), src);
    fs.writeFileSync(path.join(tmp, 'delaunator_stub.js'),
        'export default class { constructor(points){ this.triangles = globalThis.__TRI__(points); } }\n');
    globalThis.document = { createElement: () => ({ style: {}, width: 0, height: 0,
        getContext: () => ({ clearRect() {}, drawImage() {}, getImageData() { throw new Error('no DOM'); }, putImageData() {} }) }) };
    if (typeof globalThis.ImageData === 'undefined')
        globalThis.ImageData = class { constructor(data, w, h) { this.data = data; this.width = w; this.height = h; } };
    const M = await import(pathToFileURL(path.join(tmp, 'Homography.js')).href);
    return { M, Homography: M.Homography, cleanup: () => fs.rmdirSync(tmp, { recursive: true }) };
}
