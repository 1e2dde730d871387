// png.mjs -- minimal PNG reader/writer (8-bit RGBA / RGB / grey / palette, non-interlaced) on Node's zlib, so the
// drop-in class can be used from a command line without the `canvas` / `pngjs` packages an older reference snapshot
// relied on (SURVEY.md §8f-4: "I/O either side of the path").  Output of decode() is ImageData-shaped.
import zlib from 'zlib';

const SIG = Buffer.from([137, 80, 78, 71, 13, 10, 26, 10]);

let crcTable = null;
function crc32(buf) {
    if (!crcTable) {
        crcTable = new Uint32Array(256);
        for (let n = 0; n < 256; n++) { let c = n; for (let k = 0; k < 8; k++) c = (c & 1) ? (0xedb88320 ^ (c >>> 1)) : (c >>> 1); crcTable[n] = c >>> 0; }
    }
    let c = 0xffffffff;
    for (let i = 0; i < buf.length; i++) c = crcTable[(c ^ buf[i]) & 0xff] ^ (c >>> 8);
    return (c ^ 0xffffffff) >>> 0;
}

export function decode(buffer) {
    const b = Buffer.isBuffer(buffer) ? buffer : Buffer.from(buffer);
    if (b.length < 8 || !b.slice(0, 8).equals(SIG)) throw ('png: not a PNG file');
    let pos = 8, width = 0, height = 0, depth = 0, ctype = 0, interlace = 0, palette = null, trns = null;
    const idat = [];
    while (pos + 8 <= b.length) {
        const len = b.readUInt32BE(pos), type = b.toString('latin1', pos + 4, pos + 8), data = b.slice(pos + 8, pos + 8 + len);
        pos += 12 + len;
        if (type === 'IHDR') { width = data.readUInt32BE(0); height = data.readUInt32BE(4); depth = data[8]; ctype = data[9]; interlace = data[12]; }
        else if (type === 'PLTE') palette = data;
        else if (type === 'tRNS') trns = data;
        else if (type === 'IDAT') idat.push(data);
        else if (type === 'IEND') break;
    }
    if (depth !== 8 || interlace !== 0) throw (`png: only 8-bit non-interlaced images are supported (depth ${depth}, interlace ${interlace})`);
    const channels = { 0: 1, 2: 3, 3: 1, 4: 2, 6: 4 }[ctype];
    if (!channels) throw (`png: unsupported colour type ${ctype}`);
    const raw = zlib.inflateSync(Buffer.concat(idat));
    const stride = width * channels, bpp = channels;
    const rows = Buffer.alloc(stride * height);
    let prev = Buffer.alloc(stride);
    for (let y = 0; y < height; y++) {
        const ft = raw[y * (stride + 1)], line = raw.slice(y * (stride + 1) + 1, (y + 1) * (stride + 1)), cur = rows.slice(y * stride, (y + 1) * stride);
        for (let i = 0; i < stride; i++) {
            const a = i >= bpp ? cur[i - bpp] : 0, up = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            let p;
            switch (ft) {
                case 0: p = 0; break;
                case 1: p = a; break;
                case 2: p = up; break;
                case 3: p = (a + up) >> 1; break;
                case 4: { const pa = Math.abs(up - c), pb = Math.abs(a - c), pc = Math.abs(a + up - 2 * c); p = (pa <= pb && pa <= pc) ? a : (pb <= pc ? up : c); break; }
                default: throw (`png: bad filter type ${ft}`);
            }
            cur[i] = (line[i] + p) & 0xff;
        }
        prev = cur;
    }
    const data = new Uint8ClampedArray(width * height * 4);
    for (let i = 0, n = width * height; i < n; i++) {
        let r, g, bl, a = 255;
        if (ctype === 6) { r = rows[4 * i]; g = rows[4 * i + 1]; bl = rows[4 * i + 2]; a = rows[4 * i + 3]; }
        else if (ctype === 2) { r = rows[3 * i]; g = rows[3 * i + 1]; bl = rows[3 * i + 2]; }
        else if (ctype === 0) { r = g = bl = rows[i]; }
        else if (ctype === 4) { r = g = bl = rows[2 * i]; a = rows[2 * i + 1]; }
        else { const k = rows[i]; r = palette[3 * k]; g = palette[3 * k + 1]; bl = palette[3 * k + 2]; a = (trns && k < trns.length) ? trns[k] : 255; }
        data[4 * i] = r; data[4 * i + 1] = g; data[4 * i + 2] = bl; data[4 * i + 3] = a;
    }
    return { data, width, height };
}

function chunk(type, data) {
    const head = Buffer.alloc(8);
    head.writeUInt32BE(data.length, 0); head.write(type, 4, 'latin1');
    const crc = Buffer.alloc(4);
    crc.writeUInt32BE(crc32(Buffer.concat([head.slice(4), data])), 0);
    return Buffer.concat([head, data, crc]);
}

/** Encodes an ImageData-shaped {data, width, height} as an 8-bit RGBA PNG (filter 0, zlib default level). */
export function encode(image) {
    const { data, width, height } = image;
    const ihdr = Buffer.alloc(13);
    ihdr.writeUInt32BE(width, 0); ihdr.writeUInt32BE(height, 4); ihdr[8] = 8; ihdr[9] = 6;
    const raw = Buffer.alloc((width * 4 + 1) * height);
    for (let y = 0; y < height; y++) Buffer.from(data.buffer, data.byteOffset + y * width * 4, width * 4).copy(raw, y * (width * 4 + 1) + 1);
    return Buffer.concat([SIG, chunk('IHDR', ihdr), chunk('IDAT', zlib.deflateSync(raw)), chunk('IEND', Buffer.alloc(0))]);
}
