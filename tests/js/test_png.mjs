// PNG codec (no GPU): decode the reference's fixture, re-encode, decode again; `--gpu`: the reference's known-answer pair
// (test/nodeTest.js flow: testImgLogoBlack.png -> transformedImage.png) through the drop-in class on the GPU, byte for byte.
import fs from 'fs';
import path from 'path';
import { fileURLToPath } from 'url';
import { decode, encode } from '../../homography.js_amd/js/png.mjs';
import { Homography } from '../../homography.js_amd/js/Homography.mjs';

const HERE = path.dirname(fileURLToPath(import.meta.url));
const FIX = path.join(HERE, '..', 'golden', 'ref_fixture');
const fails = [];
const ok = (c, m) => { if (!c) fails.push(m); };
const same = (a, b) => a.length === b.length && Buffer.compare(Buffer.from(a.buffer, a.byteOffset, a.byteLength), Buffer.from(b.buffer, b.byteOffset, b.byteLength)) === 0;

const src = decode(fs.readFileSync(path.join(FIX, 'testImgLogoBlack.png')));
const want = decode(fs.readFileSync(path.join(FIX, 'transformedImage.png')));
ok(src.width === 400 && src.height === 400 && src.data.length === 640000, 'fixture input size');
ok(want.width === 400 && want.height === 200, 'fixture output size');
const again = decode(encode(src));
ok(again.width === src.width && again.height === src.height && same(again.data, src.data), 'encode/decode round trip');
let nz = 0; for (let i = 3; i < src.data.length; i += 4) if (src.data[i]) nz++;
ok(nz > 1000, 'fixture decodes to a non-empty alpha channel');

if (process.argv.includes('--gpu')) {
    const h = new Homography();                                                        // test/nodeTest.js:5-13
    h.setReferencePoints([[0, 0], [0, 1], [1, 0], [1, 1]], [[1 / 10, 1 / 2], [0, 1], [9 / 10, 1 / 2], [1, 1]]);
    h.setImage(src);
    const out = h.warp();
    ok(out.width === want.width && out.height === want.height, `known answer size ${out.width}x${out.height}`);
    let diff = 0; for (let i = 0; i < want.data.length; i += 4) if (out.data[i] !== want.data[i] || out.data[i + 1] !== want.data[i + 1] || out.data[i + 2] !== want.data[i + 2] || out.data[i + 3] !== want.data[i + 3]) diff++;
    ok(diff === 0, `${diff} of ${want.width * want.height} pixels differ from the reference's transformedImage.png`);
    h.close();
}
console.log(JSON.stringify({ failures: fails }));
process.exit(fails.length ? 1 : 0);
