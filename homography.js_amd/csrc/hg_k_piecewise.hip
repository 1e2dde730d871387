// hg_k_piecewise.hip -- inverse piecewise-affine warp: k_tri_setup, k_pw_fused (general), k_tri_spans + k_pw_rows (fast path)
// Hand-written HIP for gfx950 (MI355X / CDNA4), wave64; fp64 coordinate math with contraction off so that nearest-neighbour
// source selection is bit-identical to the reference's JS doubles.
// Citations are file:line into the reference's Homography.js (v1.8.0).  Design notes: DESIGN.md §4.
#include "hg_dev.h"
#include <type_traits>

namespace hg {

// ------------------------------------------------------------------------------------------------ k_tri_setup
// Per (frame, triangle).  Replaces _calculatePiecewiseAffineTransformMatrices :785-804, the inverseAffineMatrix loop
// :1036-1038 and the per-triangle head of fillTriangle :1113-1118.
constexpr int kBandMax = 2048;                 // candidate bands per frame the filing below can aggregate in LDS (host: self-span path only below that)

// One triangle of one frame: solves, edge equations, row range, column reach; returns false when the triangle has no rows or is irregular.
__device__ __forceinline__ bool tri_setup_one(const PwMesh &mesh, const PwFrames &fr, int f, int t, const FrameDesc &fd, TriRange &tr)
{
    const float *dp = fr.dst_pts + (size_t)f * mesh.n_pts * 2;
    float s[6], d[6];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t v = mesh.tris[3 * (size_t)t + k];
        if (v < (uint32_t)mesh.n_pts) {
            s[2 * k] = mesh.src_pts[2 * (size_t)v]; s[2 * k + 1] = mesh.src_pts[2 * (size_t)v + 1];
            d[2 * k] = dp[2 * (size_t)v];           d[2 * k + 1] = dp[2 * (size_t)v + 1];
        } else {                                   // typed-array read past the end: undefined -> NaN in the Float32Array(6)
            s[2 * k] = s[2 * k + 1] = d[2 * k] = d[2 * k + 1] = NAN;
        }
    }
    const size_t ft = (size_t)f * mesh.n_tris + t;
    float fwd[6], inv[6];
    solve_affine(s, d, fwd);
    invert_affine(fwd, inv);
#pragma unroll
    for (int k = 0; k < 6; k++) fr.fwd[ft * 6 + k] = fwd[k];
    *reinterpret_cast<float4 *>(fr.inv + ft * kInvStride) = make_float4(inv[0], inv[1], inv[2], inv[3]);
    *reinterpret_cast<float4 *>(fr.inv + ft * kInvStride + 4) = make_float4(inv[4], inv[5], 0.f, 0.f);
    const bool one_fma = affine_fusable(inv, coord_bits(fd.x_off, fd.obj_w), coord_bits(fd.y_off, fd.obj_h));     // (hg_math.h)

    Seg *sg = fr.segs + ft * 3;
    Seg a, b, c;
    define_seg(d[0], d[1], d[2], d[3], a);          // p0->p1
    define_seg(d[0], d[1], d[4], d[5], b);          // p0->p2
    define_seg(d[2], d[3], d[4], d[5], c);          // p1->p2
    sg[0] = a; sg[1] = b; sg[2] = c;

    tri_rows_tight(d[1], d[3], d[5], tr.y_min, tr.y_end);    // (without the leading row that cannot have a span, hg_math.h)
    tr.a = 0; tr.b = 0;
    const bool too_tall = (tr.y_end - (int64_t)tr.y_min) > (1 << 24);
    {   // rows that cannot write a cell are dropped from the range (hg_math.h): every consumer -- k_pw_fused, k_map_fill, the
        // span prologues of k_pw_rows<SELF> / k_pw_patch<SELF> -- walks [y_min, y_end) and none of those rows has a span
        int64_t y_first = tr.y_min, y_stop = tr.y_end;
        if (fd.obj_w > 0 && fd.obj_h > 0) clamp_rows(y_first, y_stop, fd.y_off, fd.obj_w, (int64_t)fd.obj_w * fd.obj_h);
        else y_stop = y_first;
        if (y_stop < y_first) y_stop = y_first;
        tr.y_min = (int32_t)y_first; tr.y_end = (int32_t)y_stop;
    }
    bool regular = false;
    if (tr.y_end > tr.y_min && fd.obj_w > 0 && fd.obj_h > 0) {
        // Conservative cell extent of any span of this triangle relative to its row base (y - yOff) * W:
        // intersections lie between the vertex x's (+-1 for rounding).  Absurd / non-finite input or a triangle wider
        // than the whole map (TypedArray.fill wrap-around could then straddle index 0) goes to the exact map path.
        const double x0 = d[0], x1 = d[2], x2 = d[4];
        const bool finite = fabs(x0) < 1.0e9 && fabs(x1) < 1.0e9 && fabs(x2) < 1.0e9 &&
                            fabs((double)d[1]) < 1.0e9 && fabs((double)d[3]) < 1.0e9 && fabs((double)d[5]) < 1.0e9;
        bool irregular = !finite || too_tall;
        if (!irregular) {
            const int64_t lo = (int64_t)floor(fmin(fmin(x0, x1), x2)) - 1;
            const int64_t hi = (int64_t)ceil(fmax(fmax(x0, x1), x2)) + 1;
            const int64_t len = (int64_t)fd.obj_w * fd.obj_h;
            if (hi - lo >= len) irregular = true;
            else {
                tr.a = (int32_t)floordiv64(hi - 1, fd.obj_w);
                tr.b = (int32_t)floordiv64(lo, fd.obj_w);
            }
        }
        if (irregular) flag_frame(fr, f, FRAME_IRREGULAR);
        regular = !irregular;
        if (regular) fr.trix[ft] = make_int2((int)floor(fmin(fmin(x0, x1), x2)) - 1, (int)ceil(fmax(fmax(x0, x1), x2)) + 1);
    }
    if (!regular) fr.trix[ft] = make_int2(0, -1);
    fr.trir[ft] = tr;
    // a triangle with rows whose sums of :1383 may round twice: the frame keeps the two-rounding pixel body (PwFrames::two_round; plain
    // store -- every writer of a step writes the same number)
    if (fr.two_round && !one_fma && tr.y_end > tr.y_min) fr.two_round[f] = fr.gen;      // (nullptr: the single-frame redo / reference-state set-ups, whose consumers never ask)
    return regular;
}

template <bool BANDS>      // BANDS: also file the triangle under its candidate bands (16 KB of LDS counters: only that instantiation carries them)
__global__ __launch_bounds__(256) void k_tri_setup(PwMesh mesh, PwFrames fr)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    const FrameDesc fd = fr.frames[f];
    TriRange tr = TriRange{0, 0, 0, 0};
    const bool regular = t < mesh.n_tris && tri_setup_one(mesh, fr, f, t, fd, tr);
    if constexpr (BANDS) {

    // Self-span path, large meshes: the triangle is filed under every band of output rows one of its spans can reach -- rows
    // (y - yOff) + b .. + a for its rows y, and objH further down for spans whose fill() index wrapped (image 1).  Slots are handed
    // out through LDS (two passes over the thread's bands: count, then file), one global atomic per (workgroup, band): a per-entry
    // device-scope atomic on a few hundred counters cost 55 us on C5 and 150 us on 64 frames of 3200 triangles.
    __shared__ int s_cnt[kBandMax], s_base[kBandMax];
    const int nb = fr.n_bands;                              // (<= kBandMax: host)
    for (int i = threadIdx.x; i < nb; i += 256) s_cnt[i] = 0;
    __syncthreads();
    int64_t lo0 = 1, hi0 = 0, lo1 = 1, hi1 = 0;
    bool file = regular;
    if (file) {
        const int64_t H = fd.obj_h;
        lo0 = (int64_t)tr.y_min - fd.y_off + tr.b; hi0 = (int64_t)tr.y_end - 1 - fd.y_off + tr.a;
        lo1 = lo0 + H; hi1 = hi0 + H;
        lo0 = lo0 < 0 ? 0 : lo0; hi0 = hi0 > H - 1 ? H - 1 : hi0;
        lo1 = lo1 < 0 ? 0 : lo1; hi1 = hi1 > H - 1 ? H - 1 : hi1;
        if (lo1 <= hi1 && lo0 <= hi0 && lo1 <= hi0 + 1) { hi0 = hi1 > hi0 ? hi1 : hi0; lo0 = lo1 < lo0 ? lo1 : lo0; hi1 = lo1 - 1; }      // overlapping: one range
        if (tr.a < -32768 || tr.a > 32767 || tr.b < -32768 || tr.b > 32767) { flag_frame(fr, f, FRAME_IRREGULAR); file = false; }
    }
    const int4 ent = make_int4(t, tr.y_min, tr.y_end, (tr.a & 0xffff) | (int)((uint32_t)tr.b << 16));
    const int2 tx = file ? fr.trix[(size_t)f * mesh.n_tris + t] : make_int2(0, -1);      // (written by this thread above)
    const int4 ent2 = make_int4(tx.x, tx.y, 0, 0);
    // (an image-1 range that shares a band with the image-0 range files the triangle twice there: harmless, a duplicate candidate
    //  produces duplicate spans of the same id)
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        if (file) {
#pragma unroll 1
            for (int image = 0; image < 2; image++) {
                const int64_t lo = image ? lo1 : lo0, hi = image ? hi1 : hi0;
                if (lo > hi) continue;
                for (int band = (int)(lo >> fr.band_rows_log2); band <= (int)(hi >> fr.band_rows_log2); band++) {
                    const int local = atomicAdd(&s_cnt[band], 1);
                    if (pass == 1) {
                        const int slot = s_base[band] + local;
                        if (slot < fr.band_cap) {
                            int4 *dst = fr.band_ent + (((size_t)f * fr.n_bands + band) * fr.band_cap + slot) * 2;
                            dst[0] = ent; dst[1] = ent2;
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (pass == 0) {
            for (int i = threadIdx.x; i < nb; i += 256) {
                const int n = s_cnt[i];
                if (n > 0) {
                    const int base = atomicAdd(&fr.band_cnt[(size_t)f * fr.band_stride + i], n);
                    s_base[i] = base;
                    if (base + n > fr.band_cap) flag_frame(fr, f, FRAME_LDS_OVERFLOW);
                }
                s_cnt[i] = 0;
            }
            __syncthreads();
        }
    }
    }   // BANDS
}

// ------------------------------------------------------------------------------------------------ k_pw_fused
// One workgroup (4 waves) per output row of one frame.
//   phase 1: every thread scans triangles; for each (triangle, source-row y) whose fillTriangle span can touch this
//            output row it evaluates predictXLimits + the flat fill() indices exactly and appends the clipped span
//            [lo,hi) x id to an LDS list;
//   phase 2: each wave walks 256-pixel windows of the row; the spans overlapping a window are found with one ballot
//            per 64 spans, and each lane keeps max(id) over the spans covering its 4 pixels ("last writer wins" of
//            the sequential fill loop :852-858 == largest id); then the pixel loop body :1044-1053.
__global__ __launch_bounds__(256) void k_pw_fused(PwMesh mesh, PwFrames fr, uint8_t *__restrict__ out, int16_t *__restrict__ map_out)
{
    const int f = blockIdx.y;
    const FrameDesc fd = fr.frames[f];
    const int r = blockIdx.x;
    if (r >= fd.obj_h || fd.obj_w <= 0) return;
    if (fr.status[f] & FRAME_IRREGULAR) return;      // written by k_tri_setup (previous kernel on this stream)

    __shared__ int s_lo[kRowSpanCap], s_hi[kRowSpanCap], s_id[kRowSpanCap];
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();

    const int T = mesh.n_tris, W = fd.obj_w;
    const int64_t len = (int64_t)W * fd.obj_h;
    const int64_t row0 = (int64_t)r * W, row1 = row0 + W;
    const TriRange *__restrict__ trir = fr.trir + (size_t)f * T;
    const Seg *__restrict__ segs = fr.segs + (size_t)f * T * 3;

    for (int t = threadIdx.x; t < T; t += 256) {
        const TriRange tr = trir[t];
        if (tr.y_end <= tr.y_min) continue;
#pragma unroll 1
        for (int image = 0; image < 2; image++) {    // 0: indices >= 0;  1: negative indices wrapped by +len (= +objH rows)
            const int64_t shift = image ? fd.obj_h : 0;
            int64_t ylo = (int64_t)r - tr.a - shift + fd.y_off, yhi = (int64_t)r - tr.b - shift + fd.y_off;
            if (ylo < tr.y_min) ylo = tr.y_min;
            if (yhi > (int64_t)tr.y_end - 1) yhi = (int64_t)tr.y_end - 1;
#pragma unroll 1
            for (int64_t y = ylo; y <= yhi; y++) {
                int64_t k, fin;
                span_cells(segs + 3 * (size_t)t, (double)y, (double)fd.y_off, (double)W, len, k, fin);
                if (k < row0) k = row0;
                if (fin > row1) fin = row1;
                if (k < fin) {
                    const int slot = atomicAdd(&s_cnt, 1);
                    if (slot < kRowSpanCap) { s_lo[slot] = (int)(k - row0); s_hi[slot] = (int)(fin - row0); s_id[slot] = t; }
                }
            }
        }
    }
    __syncthreads();
    const int cnt = s_cnt;
    if (cnt > kRowSpanCap) {                         // frame is redone through the materialised-map path by the host
        if (threadIdx.x == 0) flag_frame(fr, f, FRAME_LDS_OVERFLOW);
        return;
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwin = (W + 255) >> 8;
    const float *__restrict__ invm = fr.inv + (size_t)f * T * kInvStride;
    const uint32_t *__restrict__ img32 = reinterpret_cast<const uint32_t *>(frame_img(mesh, f));
    const int64_t n_src_px = (int64_t)mesh.W * mesh.H;
    uint32_t *__restrict__ orow = reinterpret_cast<uint32_t *>(out + fd.out_off) + row0;
    const bool vec_ok = ((W & 3) == 0) && ((fd.out_off & 15) == 0);
    const double y = (double)(r + fd.y_off);
    const double bx0 = (double)mesh.min_src_x, bx1 = (double)mesh.W + (double)mesh.min_src_x;    // :1047
    const double by0 = (double)mesh.min_src_y, by1 = (double)mesh.H + (double)mesh.min_src_y;

    for (int w = wave; w < nwin; w += 4) {
        const int c0 = w << 8, cq = c0 + (lane << 2);
        int tid[4] = { -1, -1, -1, -1 };
        for (int j = 0; j < cnt; j += 64) {
            const int idx = j + lane;
            int lo = 0x7fffffff, hi = 0;
            if (idx < cnt) { lo = s_lo[idx]; hi = s_hi[idx]; }
            unsigned long long mask = __ballot(lo < c0 + 256 && hi > c0);
            while (mask) {
                const int b = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int sl = s_lo[j + b], id = s_id[j + b];
                const unsigned span = (unsigned)(s_hi[j + b] - sl);
                const int d = cq - sl;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if ((unsigned)(d + k) < span) tid[k] = max(tid[k], id);
            }
        }
        if (cq < W) {
            uint32_t px[4];
            MatCache mc; mc.id = -1;
#pragma unroll
            for (int k = 0; k < 4; k++)
                px[k] = pw_pixel(tid[k], cq + k + fd.x_off, y, mc, invm, img32, n_src_px, mesh.W, mesh.H, bx0, bx1, by0, by1);
            store_quad(orow, cq, W, vec_ok, px);
            if (map_out) {
#pragma unroll
                for (int k = 0; k < 4; k++) if (cq + k < W) map_out[fd.map_off + row0 + cq + k] = (int16_t)tid[k];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ fast path: k_tri_spans + k_pw_rows
// Two kernels per batch of frames replace _calculatePiecewiseAffineTransformMatrices (:785-804),
// _buildInverseTrianglesCorrespondencesMatrix (:845-861), the inverseAffineMatrix loop (:1036-1038) and the pixel loop
// (:1042-1056) without ever materialising the Int16 map:
//
//   k_tri_spans  triangle-major, like the reference's fill loop: one workgroup per (frame, triangle) solves the
//                triangle's forward/inverse matrices, then one thread per source row y of fillTriangle (:1120) evaluates
//                predictXLimits + the two flat fill() indices exactly (TypedArray.fill semantics incl. negative-index
//                wrap) and appends the covered cells, cut at output-row boundaries, to that OUTPUT row's span list
//                {lo, hi, triangle id, inverse matrix as 6 f32} (32 bytes, global memory, atomic slot counter per row).
//                Exact for any input; the only limit is the per-row list capacity (overflow -> frame redone via the map).
//   k_pw_rows    one workgroup per group of 4 output rows (or per row for dense meshes): loads the rows' lists into LDS
//                (matrix widened to f64 and specialised to the row: {m0, m2*y, m4, m1, m3*y, m5}; m2*y and m3*y are the
//                separately rounded products of :1383-1384), then each wave walks the 256-pixel windows of its row: spans
//                overlapping the window are found with one ballot per 64 spans and every lane keeps the LARGEST covering
//                id per pixel (== the sequential overwrite order of :852-858), then the pixel body: 1 fma + 1 add per
//                coordinate in fp64, Math.round and the bounds test :1047 through two round-toward-minus-infinity adds
//                per coordinate (see round_x8), one buffer load whose hardware range check returns 0 outside the RGBA
//                array (the JS `undefined` -> 0 case), coalesced non-temporal stores.
// Requirements checked by pw_fast_ok(): n_tris <= 32767 (ids == their Int16 value), obj_w <= 65535, source < 2^31 bytes,
// |min_src_x/y| < 2^22.


template <bool COMPACT>
__global__ __launch_bounds__(256) void k_tri_spans(PwMesh mesh, PwFrames fr, RowLists rl)
{
    const int t = blockIdx.x, f = blockIdx.y;
    const FrameDesc fd = fr.frames[f];
    const float *dp = fr.dst_pts + (size_t)f * mesh.n_pts * 2;
    float s[6], d[6];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t v = mesh.tris[3 * (size_t)t + k];
        if (v < (uint32_t)mesh.n_pts) {
            s[2 * k] = mesh.src_pts[2 * (size_t)v]; s[2 * k + 1] = mesh.src_pts[2 * (size_t)v + 1];
            d[2 * k] = dp[2 * (size_t)v];           d[2 * k + 1] = dp[2 * (size_t)v + 1];
        } else {
            s[2 * k] = s[2 * k + 1] = d[2 * k] = d[2 * k + 1] = NAN;
        }
    }
    float fwd[6], inv[6];
    solve_affine(s, d, fwd);                       // every thread redundantly: cheaper than a broadcast through LDS
    invert_affine(fwd, inv);
    Seg seg[3];
    define_seg(d[0], d[1], d[2], d[3], seg[0]);     // p0->p1
    define_seg(d[0], d[1], d[4], d[5], seg[1]);     // p0->p2
    define_seg(d[2], d[3], d[4], d[5], seg[2]);     // p1->p2
    int32_t y_min, y_end;
    tri_rows(d[1], d[3], d[5], y_min, y_end);
    const size_t ft = (size_t)f * mesh.n_tris + t;
    if (threadIdx.x == 0) {                        // taps + inputs of the map path
#pragma unroll
        for (int k = 0; k < 6; k++) fr.fwd[ft * 6 + k] = fwd[k];
        *reinterpret_cast<float4 *>(fr.inv + ft * kInvStride) = make_float4(inv[0], inv[1], inv[2], inv[3]);
        *reinterpret_cast<float4 *>(fr.inv + ft * kInvStride + 4) = make_float4(inv[4], inv[5], 0.f, 0.f);
        fr.segs[ft * 3] = seg[0]; fr.segs[ft * 3 + 1] = seg[1]; fr.segs[ft * 3 + 2] = seg[2];
        TriRange tr; tr.y_min = y_min; tr.y_end = y_end; tr.a = 0; tr.b = 0;
        fr.trir[ft] = tr;
    }
    const int W = fd.obj_w;
    if (W <= 0 || fd.obj_h <= 0) return;
    const int64_t len = (int64_t)W * fd.obj_h;
    int32_t *__restrict__ rowcnt = rl.cnt + (size_t)f * rl.row_stride;
    const size_t ent0 = (size_t)f * rl.row_stride * rl.cap;
    int64_t y_first = y_min, y_stop = y_end;
    clamp_rows(y_first, y_stop, fd.y_off, W, len);           // rows that cannot write a cell are skipped (hg_math.h)
    for (int64_t y = y_first + threadIdx.x; y < y_stop; y += blockDim.x) {
        int64_t k, fin;
        span_cells(seg, (double)y, (double)fd.y_off, (double)W, len, k, fin);
        if (k >= fin) continue;
        // usual case: the span sits in output row (y - yOff) (+objH when it wrapped); otherwise divide
        int64_t r = y - fd.y_off;
        if (r < 0) r += fd.obj_h;
        if (r < 0 || r >= fd.obj_h || k < r * W || k >= (r + 1) * W) r = k / W;
        for (; r * W < fin; r++) {
            const int64_t lo = (k > r * W ? k : r * W) - r * W, hi = (fin < (r + 1) * W ? fin : (r + 1) * W) - r * W;
            const int slot = atomicAdd(&rowcnt[r], 1);
            if (slot < rl.cap) {
                const size_t idx = ent0 + (size_t)r * rl.cap + slot;
                const uint32_t lh = (uint32_t)lo | ((uint32_t)hi << 16);
                if (COMPACT) static_cast<uint2 *>(rl.ent)[idx] = make_uint2(lh, (uint32_t)t);
                else {
                    uint4 *dst = static_cast<uint4 *>(rl.ent) + 2 * idx;
                    dst[0] = make_uint4(lh, (uint32_t)t, __float_as_uint(inv[0]), __float_as_uint(inv[1]));
                    dst[1] = make_uint4(__float_as_uint(inv[2]), __float_as_uint(inv[3]), __float_as_uint(inv[4]), __float_as_uint(inv[5]));
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------ k_tri_spans_grouped (round 3)
// The same spans into the same row lists, for frame sets with thousands of (frame, triangle) pairs.  In k_tri_spans every wave of
// a triangle's workgroup repeats that triangle's solves -- 15 fp64 divisions, ~500 dependent instructions, 2000 clocks of a SIMD
// whichever lanes are useful: at 128 threads per triangle that was 21 of the kernel's 42 us on C3 and all but a few of its
// 66 us on C5 (40 000 triangles of ~150 rows).  Here a workgroup takes kTriGroup consecutive triangles of a frame: lanes
// 0..15 of its first wave solve one triangle EACH (same 2000 clocks, 16 triangles), the results go through LDS, and then every
// wave takes whole triangles (wave w: w, w + 8) with one lane per row as before, its triangle's records read from LDS as
// wave-uniform values.  The 8 waves per workgroup keep ~6 waves per SIMD in flight on C3 for the returning slot atomics.
constexpr int kTriGroupThreads = 512;

template <bool COMPACT, int kTriGroup>       // kTriGroup: 16, or 64 (a full wave of solves) for dense meshes
__global__ __launch_bounds__(kTriGroupThreads) void k_tri_spans_grouped(PwMesh mesh, PwFrames fr, RowLists rl)
{
    __shared__ Seg s_seg[kTriGroup][3];
    __shared__ float s_inv[kTriGroup][8];
    __shared__ long long s_rows[kTriGroup][2];                  // clamped row range [y_first, y_stop)
    const int t0 = blockIdx.x * kTriGroup, f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FrameDesc fd = fr.frames[f];
    const int W = fd.obj_w;
    const int64_t len = (int64_t)W * fd.obj_h;
    if (tid < kTriGroup && t0 + tid < mesh.n_tris) {
        const int t = t0 + tid;
        const float *dp = fr.dst_pts + (size_t)f * mesh.n_pts * 2;
        float s[6], d[6];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint32_t v = mesh.tris[3 * (size_t)t + k];
            if (v < (uint32_t)mesh.n_pts) {
                s[2 * k] = mesh.src_pts[2 * (size_t)v]; s[2 * k + 1] = mesh.src_pts[2 * (size_t)v + 1];
                d[2 * k] = dp[2 * (size_t)v];           d[2 * k + 1] = dp[2 * (size_t)v + 1];
            } else {
                s[2 * k] = s[2 * k + 1] = d[2 * k] = d[2 * k + 1] = NAN;
            }
        }
        float fwd[6], inv[6];
        solve_affine(s, d, fwd); invert_affine(fwd, inv);
        Seg seg[3];
        define_seg(d[0], d[1], d[2], d[3], seg[0]);     // p0->p1
        define_seg(d[0], d[1], d[4], d[5], seg[1]);     // p0->p2
        define_seg(d[2], d[3], d[4], d[5], seg[2]);     // p1->p2
        int32_t y_min, y_end;
        tri_rows(d[1], d[3], d[5], y_min, y_end);
        const size_t ft = (size_t)f * mesh.n_tris + t;
#pragma unroll
        for (int k = 0; k < 6; k++) fr.fwd[ft * 6 + k] = fwd[k];                                 // taps + inputs of the map path
        *reinterpret_cast<float4 *>(fr.inv + ft * kInvStride) = make_float4(inv[0], inv[1], inv[2], inv[3]);
        *reinterpret_cast<float4 *>(fr.inv + ft * kInvStride + 4) = make_float4(inv[4], inv[5], 0.f, 0.f);
        fr.segs[ft * 3] = seg[0]; fr.segs[ft * 3 + 1] = seg[1]; fr.segs[ft * 3 + 2] = seg[2];
        TriRange tr; tr.y_min = y_min; tr.y_end = y_end; tr.a = 0; tr.b = 0;
        fr.trir[ft] = tr;
        int64_t y_first = y_min, y_stop = y_end;
        if (W > 0 && fd.obj_h > 0) clamp_rows(y_first, y_stop, fd.y_off, W, len);               // rows that cannot write a cell are skipped (hg_math.h)
        else y_stop = y_first;
        s_seg[tid][0] = seg[0]; s_seg[tid][1] = seg[1]; s_seg[tid][2] = seg[2];
#pragma unroll
        for (int k = 0; k < 6; k++) s_inv[tid][k] = inv[k];
        s_rows[tid][0] = y_first; s_rows[tid][1] = y_stop;
    }
    __syncthreads();
    if (W <= 0 || fd.obj_h <= 0) return;
    int32_t *__restrict__ rowcnt = rl.cnt + (size_t)f * rl.row_stride;
    const size_t ent0 = (size_t)f * rl.row_stride * rl.cap;
    const int nw = blockDim.x >> 6;
#pragma unroll 1
    for (int j = wave; j < kTriGroup && t0 + j < mesh.n_tris; j += nw) {
        const int t = t0 + j;
        Seg seg[3];
#pragma unroll
        for (int e = 0; e < 3; e++) {
            seg[e].m = sgpr_f64(s_seg[j][e].m); seg[e].b = sgpr_f64(s_seg[j][e].b);
            seg[e].minY = sgpr_f64(s_seg[j][e].minY); seg[e].maxY = sgpr_f64(s_seg[j][e].maxY);
        }
        float inv[6];
#pragma unroll
        for (int k = 0; k < 6; k++) inv[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_inv[j][k])));
        const int64_t y_first = s_rows[j][0], y_stop = s_rows[j][1];
        for (int64_t y = y_first + lane; y < y_stop; y += 64) {
            int64_t k, fin;
            span_cells(seg, (double)y, (double)fd.y_off, (double)W, len, k, fin);
            if (k >= fin) continue;
            // usual case: the span sits in output row (y - yOff) (+objH when it wrapped); otherwise divide
            int64_t r = y - fd.y_off;
            if (r < 0) r += fd.obj_h;
            if (r < 0 || r >= fd.obj_h || k < r * W || k >= (r + 1) * W) r = k / W;
            for (; r * W < fin; r++) {
                const int64_t lo = (k > r * W ? k : r * W) - r * W, hi = (fin < (r + 1) * W ? fin : (r + 1) * W) - r * W;
                const int slot = atomicAdd(&rowcnt[r], 1);
                if (slot < rl.cap) {
                    const size_t idx = ent0 + (size_t)r * rl.cap + slot;
                    const uint32_t lh = (uint32_t)lo | ((uint32_t)hi << 16);
                    if (COMPACT) static_cast<uint2 *>(rl.ent)[idx] = make_uint2(lh, (uint32_t)t);
                    else {
                        uint4 *dst = static_cast<uint4 *>(rl.ent) + 2 * idx;
                        dst[0] = make_uint4(lh, (uint32_t)t, __float_as_uint(inv[0]), __float_as_uint(inv[1]));
                        dst[1] = make_uint4(__float_as_uint(inv[2]), __float_as_uint(inv[3]), __float_as_uint(inv[4]), __float_as_uint(inv[5]));
                    }
                }
            }
        }
    }
}


template <int CAP, bool MAP, int PH, bool COMPACT, bool HIB, int SELF>      // SELF: 0 row lists, 1 own spans
__device__ __forceinline__ void pw_rows_body(const PwMesh &mesh, const PwFrames &fr, const RowLists &rl, uint8_t *__restrict__ out,
                                             int16_t *__restrict__ map_out, int groups_per_xcd, int rows_per_group,
                                             int32_t *__restrict__ status_next)
{
    // A workgroup owns rows_per_group consecutive output rows: kRowGroup = 4 when the host expects short span lists, 1
    // for dense meshes (there, rows that share source lines should run side by side in different workgroups).  1-D grid decoded so that XCD x (= block id % number of XCCs of the device -- hipDeviceAttributeNumberOfXccs, hg_create --, the observed
    // dispatch order; speed only, never correctness) walks a contiguous band of rows of one frame: vertically adjacent
    // output rows share source cache lines, which then stay in that XCD's L2 instead of being fetched by up to 8 of them.
    const int bid = blockIdx.x, xcd = bid & ((1 << fr.xcc_log2) - 1);
    const int bi = bid >> fr.xcc_log2;
    int f, gi;                                              // frame and row group of it (frame_group, hg_dev.h: bands, rotating bands or dealt sub-bands)
    if (!frame_group(fr, xcd, bi, groups_per_xcd, f, gi)) return;
    // (fr.xcc_rotate: the band an XCD takes rotates with the frame.  Where rows differ in cost -- C4's face mesh fills the middle
    //  bands and leaves the top and bottom ones nearly empty -- a fixed band per XCD hands the same XCD the expensive band of EVERY
    //  frame: C4 0.234 -> 0.220 ms.  Where they do not and the frames share one source, the fixed band is what keeps that band's
    //  source rows in the XCD's L2 from frame to frame: C3 0.551 fixed, 0.607 rotating.  The host decides, hg_api_piecewise.hip.)
    const int r0 = gi * rows_per_group;
    const FrameDesc fd = fr.frames[f];
    // housekeeping for the NEXT step (saves its memset): the next status set is cleared here, and every workgroup zeroes the
    // span counters of its rows in the OTHER of the two counter sets -- the one the previous step consumed and the next step's
    // k_tri_spans will count into (ping-pong: nobody reads it during this launch, so no ordering against this launch's readers)
    const int nthreads = (int)blockDim.x, nwaves = nthreads >> 6;       // 256 threads
    if (bid == 0 && status_next) for (int i = threadIdx.x; i < fr.n_frames; i += nthreads) status_next[i] = 0;
    // (every row of the frame's counter block, not only the rows of THIS step's window: the other set was filled under the
    //  previous step's geometry, whose frame may have been a row taller)
    if ((int)threadIdx.x < rows_per_group && r0 + (int)threadIdx.x < rl.row_stride) rl.cnt_clear[(size_t)f * rl.row_stride + r0 + threadIdx.x] = 0;
    if (r0 >= fd.obj_h || fd.obj_w <= 0) return;

    __shared__ __align__(16) double s_m[CAP * 6];
    __shared__ int s_lo[CAP], s_hi[CAP], s_len[CAP], s_key[CAP];   // span start / end (window overlap test), length, key (KS)
    constexpr int RG = kRowGroup;                            // rows per group in packed mode
    static_assert(CAP >= 64 * RG, "packed mode gives each of the RG rows a 64-slot block");
    constexpr int KS = CAP * 48 <= (1 << kKeyShift) ? kKeyShift : kKeyShift + 1, KMASK = (1 << KS) - 1;   // id << KS | record offset | unsafe
    static_assert(CAP * 48 <= (1 << KS) && KS <= 15, "record offsets must fit below the 15-bit id");
    // Record offsets are multiples of 48: the low four bits of a key are free.  Bit 0 = "unsafe": 0 only when BOTH end pixels of the
    // span pass the source bounds test :1047 -- then every pixel between them does (sx, sy are monotone in x: one exact product, two
    // monotone roundings), and a window whose pixels all resolve to such spans runs the pixel body without the bounds test.
    constexpr int KADDR = KMASK & ~15;

    const int W = fd.obj_w;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform: the window loop runs on the scalar unit
    const int nwin_row = (W + 255) >> 8;
    const int w_lo = 0, nwin = nwin_row;
    const int nrows = min(rows_per_group, fd.obj_h - r0);

    // ---- span counts of the group's rows.  Packed mode (every row has at most 63 spans: the common case): all four
    // lists are loaded at once, one barrier, then wave j walks row r0 + j alone -- the list-load latency is paid once per
    // four rows and a row's span scan is a single ballot.  Otherwise the rows are taken one after the other with the
    // whole LDS (up to CAP - 1 spans) and the windows of a row are dealt to the four waves.
    // SELF (round 4, hg_kernels.h): no row lists exist; the workgroup evaluates the spans of its own rows in the prologue below
    // (4-row groups are always "packed", one-row groups use the whole LDS) and the counts come out of it.
    constexpr int kCandCap = SELF ? 256 : 1;
    __shared__ int s_cnt[RG], s_ncand;
    __shared__ int s_cand_tn[kCandCap], s_cand_y[kCandCap];      // candidate: triangle | rows << 16, first source row
    int cnts[RG], cmax = 0;
    if (!SELF) {
        const int32_t *cntp = rl.cnt + (size_t)f * rl.row_stride + r0;
#pragma unroll
        for (int j = 0; j < RG; j++) { cnts[j] = j < nrows ? cntp[j] : 0; cmax = max(cmax, cnts[j]); }
        if (cmax > rl.cap || cmax > CAP - 1) {
            if (threadIdx.x == 0) flag_frame(fr, f, FRAME_LDS_OVERFLOW);
            return;
        }
    }
    const bool packed = SELF ? rows_per_group == RG : (rows_per_group == RG && __builtin_amdgcn_readfirstlane(cmax) <= 63);

    // Source: buffer of W*H records of 4 bytes, gathered by PIXEL index (`buffer_load_dword ... idxen`: one v_mad per address instead of a
    // mad and a shift, EXPERIMENTS.md R6.9): an index at or beyond its end (and the -1 of rejected pixels) returns 0 from the hardware
    // range check == the JS `undefined` -> 0 of :1051.
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(frame_img(mesh, f)), 4, mesh.W * mesh.H, 0x00020000);   // (records of 4 bytes: gathers by pixel index, hg_struct_load_u32)
    // :1047 on h = RTN(s + 0.5):  minSrcX <= sx < W + minSrcX  <=>  minSrcX + 0.5 <= hx < W + minSrcX + 0.5, same for y.
    // All four are tested on the doubles: the rounded coordinates are only 32-bit (a source row of 300 * 2^24 must be
    // rejected, not wrapped back into the image); what the range check of the buffer load still provides is the `undefined`
    // -> 0 of flat indices that pass :1047 and yet fall outside the array (sx in [W - 0.5, W) on the last row).
    const double bx_lo = sgpr_f64((double)mesh.min_src_x + 0.5), bx_hi = sgpr_f64((double)mesh.W + (double)mesh.min_src_x + 0.5);
    const double by_lo = sgpr_f64((double)mesh.min_src_y + 0.5), by_hi = sgpr_f64((double)mesh.H + (double)mesh.min_src_y + 0.5);
    // HIB (host: hi_bounds_ok): the same four tests as two 32-bit compares on the high dwords of h (hg_dev.h)
    const HiBounds hb = make_hi_bounds((double)mesh.min_src_x + 0.5, (double)mesh.W + (double)mesh.min_src_x + 0.5,
                                       (double)mesh.min_src_y + 0.5, (double)mesh.H + (double)mesh.min_src_y + 0.5);
    // 1 unless both end pixels lo, hi - 1 of a span with record {m0, m2*y, m4, m1, m3*y, m5} are inside the source window, computed as
    // the pixel body computes them (same fma, same rounding, same compares)
    const bool flag_spans = fr.safe_spans != 0;             // (wave-uniform; host: by the rows' span density)
    // One fma per coordinate for the whole frame: (m0 x) + (m2 y) + m4 of :1383 rounds twice, but where m0 x + m2 y and m2 y + m4 are exactly
    // representable for every pixel of the window -- k_tri_setup checked every triangle of the frame that has rows (affine_fusable, hg_math.h:
    // both are multiples of the smallest ulp of the f32 entries involved, bounded by the window) -- fma(m0, x, A) with A = (m2 y) + m4 has the same
    // bits.  The span records of such a frame hold {m0, A, m1, B}: 32 bytes of LDS per pixel instead of 48, which is what this kernel runs
    // on (EXPERIMENTS.md R5.8).  Uniform for the workgroup; the row lists (small frame sets) keep the long form.
    const bool one_fma = SELF != 0 && fr.two_round && __builtin_amdgcn_readfirstlane(fr.two_round[f]) != fr.gen;
    auto span_unsafe = [&](double m0, double m2y, double m4, double m1, double m3y, double m5, int lo, int hi) -> int {
        if (!flag_spans) return 1;
        const double xa = (double)(lo + fd.x_off), xb = (double)(hi - 1 + fd.x_off);
        double h[4] = { fma(m0, xa, m2y) + m4, fma(m1, xa, m3y) + m5, fma(m0, xb, m2y) + m4, fma(m1, xb, m3y) + m5 }, rd[4];
        round_x4(h, rd);
        const bool a = HIB ? hi_inb(hb, h[0], h[1]) : (bool)((int)(h[0] >= bx_lo) & (int)(h[0] < bx_hi) & (int)(h[1] >= by_lo) & (int)(h[1] < by_hi));
        const bool b = HIB ? hi_inb(hb, h[2], h[3]) : (bool)((int)(h[2] >= bx_lo) & (int)(h[2] < bx_hi) & (int)(h[3] >= by_lo) & (int)(h[3] < by_hi));
        return (a && b) ? 0 : 1;
    };

    const float *__restrict__ ginv = fr.inv + (size_t)f * mesh.n_tris * kInvStride;
    // row `row` of the group -> LDS slots [base, base + cnt) (+ a NaN record in slot base + nan_slot that pixels without a
    // triangle point at); threads t0, t0 + step, ... of the caller's thread set do the copying
    auto load_row = [&](int row, int cnt, int base, int nan_slot, int t0, int step) {
        const double y = (double)(r0 + row + fd.y_off);
        const size_t e0 = ((size_t)f * rl.row_stride + r0 + row) * rl.cap;
        for (int i = t0; i < cnt; i += step) {
            uint32_t lh, id;
            double m0, m1, m2, m3, m4, m5;
            if (COMPACT) {                                  // 8-byte entry; the triangle's f32 inverse matrix from the tap array (L2)
                const uint2 a = static_cast<const uint2 *>(rl.ent)[e0 + i];
                lh = a.x; id = a.y;
                const float4 ma = *reinterpret_cast<const float4 *>(ginv + (size_t)id * kInvStride);
                const float2 mb = *reinterpret_cast<const float2 *>(ginv + (size_t)id * kInvStride + 4);
                m0 = (double)ma.x; m1 = (double)ma.y; m2 = (double)ma.z; m3 = (double)ma.w; m4 = (double)mb.x; m5 = (double)mb.y;
            } else {                                        // 32-byte entry carrying the matrix
                const uint4 a = static_cast<const uint4 *>(rl.ent)[2 * (e0 + i)], b = static_cast<const uint4 *>(rl.ent)[2 * (e0 + i) + 1];
                lh = a.x; id = a.y;
                m0 = (double)__uint_as_float(a.z); m1 = (double)__uint_as_float(a.w); m2 = (double)__uint_as_float(b.x);
                m3 = (double)__uint_as_float(b.y); m4 = (double)__uint_as_float(b.z); m5 = (double)__uint_as_float(b.w);
            }
            const int elo = (int)(lh & 0xffffu), ehi = (int)(lh >> 16);
            s_lo[base + i] = elo; s_hi[base + i] = ehi; s_len[base + i] = ehi - elo;
            s_key[base + i] = ((int)id << KS) | ((base + i) * 48) | span_unsafe(m0, m2 * y, m4, m1, m3 * y, m5, elo, ehi);
            double2 *mrec = reinterpret_cast<double2 *>(s_m + (base + i) * 6);
            mrec[0] = make_double2(m0, m2 * y);              // {m0, m2*y, m1, m3*y, m4, m5}: m2*y and m3*y are the separately
            mrec[1] = make_double2(m1, m3 * y);              // rounded products of :1383-1384
            mrec[2] = make_double2(m4, m5);
        }
        if (t0 < 3) reinterpret_cast<double2 *>(s_m + (base + nan_slot) * 6)[t0] = make_double2(NAN, NAN);
    };

    // SELF prologue: the spans of this group's rows, computed here instead of being filed into row lists by a producer kernel.
    // Input: what k_tri_setup wrote per (frame, triangle) -- edge equations, inverse matrix, the clamped row range [y_min, y_end)
    // of fillTriangle's loop :1113-1120 and the bounds a, b on the output rows a row-y span can reach ((y - yOff) + b .. + a, objH
    // further down when fill() wrapped a negative index: "image" 1; the enumeration k_pw_fused has always used).
    //  (1) scan: thread t tests triangle t (16 bytes each, one round trip for T <= 256): can one of its rows write into rows
    //      r0 .. r0 + nrows - 1?  Hits are compacted into an LDS candidate list with one ballot + one LDS atomic per wave.
    //  (2) spans: 8 lanes per candidate (2 for one-row groups), one source row each: predictXLimits :1172-1197 + the two flat
    //      fill() indices :1124 exactly (span_cells, hg_math.h), cut at output-row boundaries, every piece that falls into a row
    //      of the group filed into that row's LDS block -- exactly what load_row copies from a list.  Arbitrary order, like there.
    auto self_prologue = [&]() -> bool {
        if ((int)threadIdx.x < RG) s_cnt[threadIdx.x] = 0;
        if (threadIdx.x == 0) s_ncand = 0;
        if ((int)threadIdx.x < 3 * nrows) {                 // the NaN record of every row block ("no triangle")
            const int row = threadIdx.x / 3, part = threadIdx.x - 3 * row;
            const int slot = packed ? row * 64 + 63 : CAP - 1;
            reinterpret_cast<double2 *>(s_m + slot * 6)[part] = make_double2(NAN, NAN);
        }
        __syncthreads();
        const int T = mesh.n_tris;
        const TriRange *__restrict__ trir = fr.trir + (size_t)f * T;
        // (int32 throughout: |yOff|, objH and the clamped row ranges are below 2^26 -- coordinates are limited to 2^24, hg_math.h --
        //  and a, b are cell offsets / W of a map with fewer than 2^31 cells)
        const int g_lo = r0 + fd.y_off, g_hi = r0 + nrows - 1 + fd.y_off;
        // Candidate entries carry up to `chunk` source rows (4 for 4-row groups, 1 for one-row groups: one lane per row below); a
        // triangle that reaches more rows -- window borders, spans spilling over the row end (x-offset quirk) -- files a second entry
        // for the rest, whose lanes loop if that is still more than `chunk` (a - b > 1: triangles wider than the map; rare).
        const int chunk_log2 = packed ? 2 : 0, chunk = 1 << chunk_log2;
        // (meshes beyond 1024 triangles: not the whole mesh but the entries k_tri_setup filed under this group's 64-row band, as in k_pw_patch<SELF>)
        int n_src = T;
        const int4 *__restrict__ bent = nullptr;
        if (fr.band_ent) {
            const int bandi = r0 >> fr.band_rows_log2;      // (row groups never straddle a band)
            n_src = min(fr.band_cnt[(size_t)f * fr.band_stride + bandi], fr.band_cap);
            bent = fr.band_ent + ((size_t)f * fr.n_bands + bandi) * fr.band_cap * 2;
        }
        for (int t0 = 0; t0 < n_src; t0 += nthreads) {
            const int i = t0 + (int)threadIdx.x;
            int t = i;
            TriRange tr = TriRange{0, 0, 0, 0};
            if (i < n_src) {
                if (bent) { const int4 e = bent[2 * i]; t = e.x; tr.y_min = e.y; tr.y_end = e.z; tr.a = (int16_t)(e.w & 0xffff); tr.b = e.w >> 16; }
                else tr = trir[i];
            }
            const int ylo0 = max(g_lo - tr.a, tr.y_min), n0 = min(g_hi - tr.b, tr.y_end - 1) - ylo0 + 1;
            const int ylo1 = max(g_lo - tr.a - fd.obj_h, tr.y_min), n1 = min(g_hi - tr.b - fd.obj_h, tr.y_end - 1) - ylo1 + 1;
            const unsigned long long m0 = __ballot(n0 > 0), m1 = __ballot(n1 > 0);
            if ((m0 | m1) == 0ull) continue;                // (wave-uniform)
            const unsigned long long m0b = __ballot(n0 > chunk), m1b = __ballot(n1 > chunk);
            const int c0 = __popcll(m0), c0b = __popcll(m0b), c1 = __popcll(m1), c1b = __popcll(m1b);
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_ncand, c0 + c0b + c1 + c1b);
            base = __builtin_amdgcn_readfirstlane(base);
            auto below = [&](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
            auto file = [&](int at, int y0, int n) { if (at < kCandCap) { s_cand_tn[at] = t | (min(n, 0xffff) << 16); s_cand_y[at] = y0; } };
            if (n0 > 0) file(base + below(m0), ylo0, min(n0, chunk));
            if (n0 > chunk) file(base + c0 + below(m0b), ylo0 + chunk, n0 - chunk);
            if (m1) {
                if (n1 > 0) file(base + c0 + c0b + below(m1), ylo1, min(n1, chunk));
                if (n1 > chunk) file(base + c0 + c0b + c1 + below(m1b), ylo1 + chunk, n1 - chunk);
            }
        }
        __syncthreads();
        const int nc = s_ncand;
        if (nc > kCandCap) return false;
        const int capr = packed ? 63 : CAP - 1;
        const int lpc_log2 = chunk_log2, lpc = chunk;       // lanes per candidate entry
        const double flen = (double)((int64_t)W * fd.obj_h), fW = (double)W;      // (len < 2^31: fill_frames)
        const Seg *__restrict__ gseg = fr.segs + (size_t)f * T * 3;
        for (int c0 = 0; c0 < nc; c0 += nthreads >> lpc_log2) {
            const int c = c0 + ((int)threadIdx.x >> lpc_log2), jj = threadIdx.x & (lpc - 1);
            if (c >= nc) continue;
            const int tn = s_cand_tn[c], t = tn & 0xffff, n = (int)((uint32_t)tn >> 16), ylo = s_cand_y[c];
            if (n >= 0xffff) { atomicAdd(&s_cnt[0], 1 << 20); continue; }      // (absurd reach: the count check below fails, the map path takes the frame)
            if (jj >= n) continue;
            const Seg *__restrict__ sg = gseg + (size_t)t * 3;
            const float4 ma = *reinterpret_cast<const float4 *>(ginv + (size_t)t * kInvStride);
            const float2 mb = *reinterpret_cast<const float2 *>(ginv + (size_t)t * kInvStride + 4);
            for (int j = jj; j < n; j += lpc) {
                const int ys = ylo + j;
                const double y = (double)ys;
                // predictXLimits :1172-1197, the lean form of span_cells (hg_math.h; same comparisons, same division, same results for
                // every input -- NaN compares false and leaves mn / mx alone, a division by m = 0 is computed and discarded)
                double mn = INFINITY, mx = -INFINITY;
                auto edge = [&](const Seg &q) {
                    const double x = q.m == INFINITY ? q.b : (y - q.b) / q.m;
                    const bool use = (y >= q.minY) & (y <= q.maxY) & !(q.m == 0.0);
                    mn = (use & (x < mn)) ? x : mn;
                    mx = (use & (x > mx)) ? x : mx;
                };
                // one edge at a time: with all three in flight the prologue would need 14 registers more than the pixel loop does
                // (72 instead of 56-58: 7 waves per SIMD instead of 8).  The three dependent round trips this costs are NOT what the
                // prologue's time is made of (EXPERIMENTS.md R4.8: forming the edges from the vertices in one round trip changed nothing)
#pragma unroll 1
                for (int e = 0; e < 3; e++) edge(sg[e]);
                // the two flat fill() indices :1124 = (y - yOffset) * W + Math.round(x) under TypedArray.fill's index rules: NaN -> 0,
                // trunc, negative counts from the end, clamp to [0, len] (js_fill_index, hg_math.h; maxNum / minNum absorb the NaN)
                const double base = (y - (double)fd.y_off) * fW;
                double rk = floor(mn); rk += (mn - rk >= 0.5) ? 1.0 : 0.0;      // Math.round (floor of +-Inf / NaN / |x| >= 2^52 is the value itself)
                double rf = floor(mx); rf += (mx - rf >= 0.5) ? 1.0 : 0.0;
                double vk = trunc(base + rk), vf = trunc(base + rf);
                vk = vk < 0.0 ? flen + vk : vk; vf = vf < 0.0 ? flen + vf : vf;
                const int k = (int)fmin(fmax(vk, 0.0), flen), fin = (int)fmin(fmax(vf, 0.0), flen);
                if (k >= fin) continue;
                // usual case: the span sits in output row (y - yOff) (+objH when it wrapped); otherwise divide
                int r = ys - fd.y_off;
                if (r < 0) r += fd.obj_h;
                if ((unsigned)r >= (unsigned)fd.obj_h || (unsigned)(k - r * W) >= (unsigned)W) r = k / W;
                if (r < r0) r = r0;                          // (pieces in rows above the group belong to other workgroups)
                for (; r < r0 + nrows; r++) {
                    const int rb = r * W;
                    if (rb >= fin) break;
                    const int lo = max(k - rb, 0), hi = min(fin - rb, W);
                    if (lo >= hi) continue;
                    const int row = r - r0;
                    const int slot = atomicAdd(&s_cnt[row], 1);
                    if (slot >= capr) continue;
                    const int at = (packed ? row * 64 : 0) + slot;
                    const double yr = (double)(r + fd.y_off);
                    const double m2y = (double)ma.z * yr, m3y = (double)ma.w * yr;
                    s_lo[at] = lo; s_hi[at] = hi; s_len[at] = hi - lo;
                    s_key[at] = (t << KS) | (at * 48) | span_unsafe((double)ma.x, m2y, (double)mb.x, (double)ma.y, m3y, (double)mb.y, lo, hi);
                    double2 *mrec = reinterpret_cast<double2 *>(s_m + at * 6);
                    if (one_fma) {                                               // {m0, A, m1, B}
                        mrec[0] = make_double2((double)ma.x, m2y + (double)mb.x);
                        mrec[1] = make_double2((double)ma.y, m3y + (double)mb.y);
                    } else {                                                     // {m0, m2*y, m1, m3*y, m4, m5}, see load_row
                        mrec[0] = make_double2((double)ma.x, m2y);
                        mrec[1] = make_double2((double)ma.y, m3y);
                        mrec[2] = make_double2((double)mb.x, (double)mb.y);
                    }
                }
            }
        }
        __syncthreads();
        bool ok = true;
#pragma unroll
        for (int j = 0; j < RG; j++) { cnts[j] = j < nrows ? s_cnt[j] : 0; ok = ok && cnts[j] <= capr; }
        return ok;
    };

    // All 256-pixel windows w0, w0 + wstep, ... of one row whose spans sit in LDS slots [base, base + cnt).
    // PH windows per phase: the gathers of PH windows are issued (each right after its window is resolved, so they overlap
    // the next window's arithmetic), then the PH x 4 stores.  On gfx9 loads and stores share one in-order counter (vmcnt): with
    // one window per phase every wait for gather data also waits for the previous window's write acknowledgements, and only 4
    // requests per wave are ever in flight.  A streaming copy in the same instruction forms (tools/calib_fetch: 4 loads + 4
    // stores per step 4.7 TB/s, 16 + 16 per step 5.7 TB/s) shows what that costs once the source comes from HBM.
    auto do_row = [&](int row, int cnt, int base, int nan_slot, int w0, int wstep) {
        // "no triangle": smaller than every real key, its low bits address the NaN record
        const int nan_key = (int)0x80000000u | ((base + nan_slot) * 48) | 1;      // (unsafe: its pixels must come out as offset 0xffffffff)
        const int r = r0 + row;
        const int64_t row_px = (int64_t)r * W;
        // Output row: raw buffer of 4*W bytes, so the ragged last window needs no per-pixel guard (stores past the row
        // end are dropped by the hardware range check).
        const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(out + fd.out_off + row_px * 4, 0, W * 4, 0x00020000);
        // (16-byte stores need 16-byte aligned addresses and must not straddle the row end: the range check drops a store whole)
        const bool vec_zero = ((W & 3) == 0) && (((fd.out_off + (uint64_t)row_px * 4) & 15) == 0);
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u zero4 = { 0u, 0u, 0u, 0u };
        // Rows of up to 64 spans (every row of a packed group): lane i keeps span i in registers for the whole row, and the window loop
        // takes the spans that reach a window out of those lanes with v_readlane -- no LDS round trip per (window, span), which was a
        // dependent ds_read + wait in front of every span's four compares (C4: 4.5 spans per window).
        const bool in_regs = cnt <= 64;                     // wave-uniform
        int lo_r = 0x7fffffff, hi_r = 0, key_r = 0;
        if (in_regs && lane < cnt) { lo_r = s_lo[base + lane]; hi_r = s_hi[base + lane]; key_r = s_key[base + lane]; }
        for (int wb = w0; wb < nwin; wb += wstep * PH) {
            uint32_t px[PH][4];
            bool empty[PH];                                 // wave-uniform
#pragma unroll
            for (int p = 0; p < PH; p++) {
                const int w = wb + p * wstep;               // wave-uniform
                if (w >= nwin) break;
                const int c0 = w << 8, cq = c0 + lane;      // lane l owns pixels c0 + l + 64k: every gather instruction covers
                int best[4];                                // 64 consecutive pixels and every store instruction 256 contiguous bytes
#pragma unroll
                for (int k = 0; k < 4; k++) best[k] = nan_key;
                unsigned long long any = 0ull;
                if (in_regs) {
                    unsigned long long mask = __ballot(lo_r < c0 + 256 && hi_r > c0);
                    any = mask;
                    while (mask) {
                        const int bit = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        const int lo = __builtin_amdgcn_readlane(lo_r, bit);
                        span_max4s(best, cq - lo, __builtin_amdgcn_readlane(hi_r, bit) - lo, __builtin_amdgcn_readlane(key_r, bit));
                    }
                } else for (int j = 0; j < cnt; j += 64) {
                    const int idx = j + lane;
                    int lo = 0x7fffffff, hi = 0;
                    if (idx < cnt) { lo = s_lo[base + idx]; hi = s_hi[base + idx]; }
                    unsigned long long mask = __ballot(lo < c0 + 256 && hi > c0);
                    any |= mask;
                    while (mask) {
                        const int bit = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        const int slot = base + j + bit;
                        const int d = cq - s_lo[slot];
                        span_max4(best, d, s_len[slot], s_key[slot]);   // larger id wins (== last writer of :852-858); its slot rides along
                    }
                }
                empty[p] = any == 0;                         // no span of this row reaches the window
                if (empty[p]) px[p][0] = px[p][1] = px[p][2] = px[p][3] = 0u;
                else {
                    const double xd0 = (double)(cq + fd.x_off);     // exact: integers far below 2^53
                    // Pixels are transformed and rounded STEP at a time.  With 2 or 4 windows per phase two at a time: only 8 + 8
                    // instead of 16 + 16 registers of coordinates are live at once, which brings the 2-window instantiation
                    // from 66 to 58 VGPRs (8 waves/SIMD instead of 7; C3 -1.4 %, and the 4-window one from 78 to 66).  With one
                    // window per phase (one source per frame) all four at once measured 2 % faster.
                    constexpr int STEP = PH == 1 ? 4 : 2;
                    // every pixel of the window resolved to a span whose ends are inside the source window: no bounds test, and Math.round
                    // with one add per coordinate (round_half_x4)
                    // :1383-1384 for pixels k0 .. k0 + N - 1 of the window: (m0*x) + (m2*y) + m4.  m0*x is exact in fp64 (24-bit f32 significand times
                    // an integer below 2^24), so fma(m0, x, m2*y) == RN((m0*x) + (m2*y)) bit for bit; a one_fma frame's records already hold
                    // A = (m2*y) + m4 in that place and nothing is added (the branch is uniform for the workgroup)
                    auto coords = [&](auto n_tag, int k0, double *v) {
                        constexpr int N = decltype(n_tag)::value;
                        if constexpr (SELF == 0) {          // (row lists: always the long form)
#pragma unroll
                            for (int k = 0; k < N; k++) {
                                const double2 *mrec = reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(s_m) + (best[k0 + k] & KADDR));
                                const double2 ra = mrec[0], rb = mrec[1], rc = mrec[2];
                                const double xd = xd0 + (double)((k0 + k) * 64);
                                v[2 * k] = fma(ra.x, xd, ra.y) + rc.x; v[2 * k + 1] = fma(rb.x, xd, rb.y) + rc.y;
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < N; k++) {
                                const double2 *mrec = reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(s_m) + (best[k0 + k] & KADDR));
                                const double2 ra = mrec[0], rb = mrec[1];
                                const double xd = xd0 + (double)((k0 + k) * 64);
                                v[2 * k] = fma(ra.x, xd, ra.y); v[2 * k + 1] = fma(rb.x, xd, rb.y);
                            }
                            if (!one_fma) {
#pragma unroll
                                for (int k = 0; k < N; k++) {
                                    const double2 rc = reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(s_m) + (best[k0 + k] & KADDR))[2];
                                    v[2 * k] += rc.x; v[2 * k + 1] += rc.y;
                                }
                            }
                        }
                    };
                    typedef std::integral_constant<int, 2> two_t;
                    typedef std::integral_constant<int, STEP> step_t;
                    const bool safe = flag_spans && __ballot(((best[0] | best[1] | best[2] | best[3]) & 1) != 0) == 0ull;      // wave-uniform
                    if (safe) {
#pragma unroll
                        for (int kk = 0; kk < 4; kk += 2) {
                            if (kk == 2 && c0 + 128 >= W) { px[p][2] = px[p][3] = 0u; continue; }
                            double v[4];
                            int r[4];
                            coords(two_t{}, kk, v);
                            round_half_x4(v, r);
#pragma unroll
                            for (int k = kk; k < kk + 2; k++)
                                px[p][k] = hg_struct_load_u32(src, __mul24(r[2 * (k - kk) + 1], mesh.W) + r[2 * (k - kk)], 0, 0, 0);     // :1048-1049, in pixels
                        }
                    } else
#pragma unroll
                    for (int kk = 0; kk < 4; kk += STEP) {
                        // the ragged last window of a row: a pair of 64-pixel pieces wholly past the row end is not computed at all (its
                        // stores would be dropped by the range check anyway; wave-uniform test) -- a 2170-pixel row has 2 dead pieces in 36
                        if (STEP == 2 && kk == 2 && c0 + 128 >= W) { px[p][2] = px[p][3] = 0u; continue; }
                        double h[2 * STEP], rd[2 * STEP];
                        coords(step_t{}, kk, h);
                        if constexpr (STEP == 4) round_x8(h, rd); else round_x4(h, rd);
#pragma unroll
                        for (int k = kk; k < kk + STEP; k++) {
                            const int q = 2 * (k - kk);
                            const bool inb = HIB ? hi_inb(hb, h[q], h[q + 1])
                                                 : (bool)((int)(h[q] >= bx_lo) & (int)(h[q] < bx_hi) & (int)(h[q + 1] >= by_lo) & (int)(h[q + 1] < by_hi));   // NaN fails
                            const int o = __mul24((int)dlo(rd[q + 1]), mesh.W) + (int)dlo(rd[q]);                           // :1048-1049, in pixels
                            px[p][k] = hg_struct_load_u32(src, inb ? o : -1, 0, 0, 0);                       // range-checked buffer load: outside the array -> 0
                        }
                    }
                }
                if (MAP) {                                  // parity tap (hg_get_tri_map_fused): a separate instantiation
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (cq + k * 64 < W) map_out[fd.map_off + row_px + cq + k * 64] = best[k] < 0 ? (int16_t)-1 : (int16_t)(best[k] >> KS);
                }
            }
#pragma unroll
            for (int p = 0; p < PH; p++) {
                const int w = wb + p * wstep;
                if (w >= nwin) break;
                const int cq = (w << 8) + lane;
                if (empty[p] && vec_zero) {                 // 1 KB of zeros: which lane writes which pixel does not matter -> one 16-byte store per lane
                    __builtin_amdgcn_raw_buffer_store_b128(zero4, dst, ((w << 8) + lane * 4) * 4, 0, kStoreNT);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) __builtin_amdgcn_raw_buffer_store_b32(px[p][k], dst, (cq + k * 64) * 4, 0, kStoreNT);
                }
            }
        }
    };

    // One window per wave iteration.  Measured alternatives (EXPERIMENTS.md): two windows in flight per wave (74 VGPRs, 6
    // waves/SIMD) and a phased variant (4 windows resolved, then 16 gathers, then 16 stores) are both ~4 % slower: with
    // 8 waves/SIMD the other waves already cover a window's memory latency, and reads + writes together run at ~5.2 TB/s.
    if (SELF) {
        if (!self_prologue()) {                             // more candidates / spans than the LDS holds: the host redoes the frame through the map
            if (threadIdx.x == 0) flag_frame(fr, f, FRAME_LDS_OVERFLOW);
            return;
        }
        const int row = packed ? wave : 0;
        int cnt_row = 0;
#pragma unroll
        for (int j = 0; j < RG; j++) cnt_row += cnts[j] * (row == j);
        const int cnt = __builtin_amdgcn_readfirstlane(cnt_row);
        if (row < nrows) do_row(row, cnt, packed ? wave * 64 : 0, packed ? 63 : CAP - 1, w_lo + (packed ? 0 : wave), packed ? 1 : nwaves);
        return;
    }
    const int npass = packed ? 1 : nrows;
    for (int pass = 0; pass < npass; pass++) {
        const int row = packed ? wave : pass;               // everything below is wave-uniform (scalar registers)
        const int cnt = __builtin_amdgcn_readfirstlane(cnts[0] * (row == 0) + cnts[1] * (row == 1) + cnts[2] * (row == 2) + cnts[3] * (row == 3));
        const int base = packed ? wave * 64 : 0, nan_slot = packed ? 63 : CAP - 1;
        load_row(row, cnt, base, nan_slot, packed ? lane : (int)threadIdx.x, packed ? 64 : nthreads);
        __syncthreads();
        if (row < nrows) do_row(row, cnt, base, nan_slot, w_lo + (packed ? 0 : wave), packed ? 1 : nwaves);
        if (!packed) __syncthreads();                       // the next row overwrites the records
    }
}

template <int CAP, bool MAP, int PH = 1, bool COMPACT = false, bool HIB = false, int SELF = 0>
__global__ __launch_bounds__(256) void k_pw_rows(PwMesh mesh, PwFrames fr, RowLists rl, uint8_t *__restrict__ out, int16_t *__restrict__ map_out,
                                                 int groups_per_xcd, int rows_per_group, int32_t *__restrict__ status_next)
{
    pw_rows_body<CAP, MAP, PH, COMPACT, HIB, SELF>(mesh, fr, rl, out, map_out, groups_per_xcd, rows_per_group, status_next);
}

// The same kernel held to 80 SGPRs.  A 256-thread workgroup puts one wave on each SIMD and a SIMD has 800 SGPRs, allocated in
// granules of 16 (+16): the 89-104 the compiler takes by itself admit 7 (or 6) workgroups per CU although VGPRs and LDS allow
// 8; capped, ~20 scalars move into VGPR lanes and 8 workgroups fit.  Measured on one box (round 3): with a shared,
// cache-resident source (instruction-bound) 2 windows per phase 0.555 -> 0.543 ms on C3; with one source per frame (HBM-bound) the
// extra waves LOSE 1-3 %, and C4's 4-windows-per-phase layout loses 6 % -- so only the shared-source PH = 2 layout takes it.
template <int CAP, bool MAP, int PH = 1, bool COMPACT = false, bool HIB = false, int SELF = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_pw_rows_s80(PwMesh mesh, PwFrames fr, RowLists rl, uint8_t *__restrict__ out,
                                                 int16_t *__restrict__ map_out, int groups_per_xcd, int rows_per_group, int32_t *__restrict__ status_next)
{
    pw_rows_body<CAP, MAP, PH, COMPACT, HIB, SELF>(mesh, fr, rl, out, map_out, groups_per_xcd, rows_per_group, status_next);
}

// ------------------------------------------------------------------------------------------------ launchers
// A frame set's block (a few KB .. a few hundred KB) from its page-locked staging slot to the device by a KERNEL that reads host memory: the
// copy engine's start-up latency made the stream-ordered hipMemcpyAsync of 36 KB cost 13 us of a 233-us step (EXPERIMENTS.md R4.13).
__global__ __launch_bounds__(256) void k_upload(UploadSegs sg)
{
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint2 *__restrict__ dst = static_cast<uint2 *>(sg.dst[k]);
        const uint2 *__restrict__ src = static_cast<const uint2 *>(sg.src[k]);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < sg.n8[k]; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    }
}

void launch_upload(const UploadSegs &sg, hipStream_t stream)     // up to three (dst, page-locked src, 8-byte words) segments in ONE launch
{
    const size_t n8 = std::max(sg.n8[0], std::max(sg.n8[1], sg.n8[2]));
    if (n8 == 0) return;
    const unsigned blocks = (unsigned)std::min<size_t>((n8 + 255) / 256, 128);
    hipLaunchKernelGGL(k_upload, dim3(blocks), dim3(256), 0, stream, sg);
}

void launch_tri_setup(const PwMesh &mesh, const PwFrames &fr, hipStream_t stream)
{
    if (mesh.n_tris <= 0 || fr.n_frames <= 0) return;
    dim3 grid((mesh.n_tris + 255) / 256, fr.n_frames);
    if (fr.band_ent) hipLaunchKernelGGL(k_tri_setup<true>, grid, dim3(256), 0, stream, mesh, fr);
    else             hipLaunchKernelGGL(k_tri_setup<false>, grid, dim3(256), 0, stream, mesh, fr);
}

void launch_pw_fused(const PwMesh &mesh, const PwFrames &fr, uint8_t *out, int16_t *map_out, hipStream_t stream)
{
    if (fr.n_frames <= 0 || fr.max_obj_h <= 0) return;
    dim3 grid(fr.max_obj_h, fr.n_frames);
    hipLaunchKernelGGL(k_pw_fused, grid, dim3(256), 0, stream, mesh, fr, out, map_out);
}

bool pw_fast_ok(const PwMesh &mesh, int max_obj_w)
{
    return mesh.n_tris > 0 && mesh.n_tris <= 32767 && max_obj_w <= 65535 && (int64_t)mesh.W * mesh.H * 4 < ((int64_t)1 << 31) &&
           mesh.W < (1 << 21) && mesh.H < (1 << 22) && std::abs(mesh.min_src_x) < (1 << 22) && std::abs(mesh.min_src_y) < (1 << 22) &&
           // the flat byte offset (round(sy) * W + round(sx)) * 4 of every pixel that passes :1047 stays inside 32 bits, so that
           // negative ones (negative source minimum) wrap to >= 2^31 and are dropped by the range check like the rest
           (((int64_t)mesh.H + std::abs(mesh.min_src_y) + 2) * mesh.W + std::abs(mesh.min_src_x) + 2) * 4 < ((int64_t)1 << 31);
}

void launch_tri_spans(const PwMesh &mesh, const PwFrames &fr, const RowLists &rl, hipStream_t stream)
{
    if (mesh.n_tris <= 0 || fr.n_frames <= 0) return;
    const dim3 grid(mesh.n_tris, fr.n_frames), block(fr.tri_threads == 64 ? 64 : (fr.tri_threads >= 256 ? 256 : 128));
    if (fr.tri_group) {                                      // thousands of (frame, triangle) pairs: the solves of 16 triangles on 16 lanes
        const int G = fr.tri_group >= 64 ? 64 : 16;
        const dim3 ggrid((mesh.n_tris + G - 1) / G, fr.n_frames), gblock(kTriGroupThreads);
        if (G == 64) {
            if (rl.compact) hipLaunchKernelGGL((k_tri_spans_grouped<true, 64>), ggrid, gblock, 0, stream, mesh, fr, rl);
            else            hipLaunchKernelGGL((k_tri_spans_grouped<false, 64>), ggrid, gblock, 0, stream, mesh, fr, rl);
        } else {
            if (rl.compact) hipLaunchKernelGGL((k_tri_spans_grouped<true, 16>), ggrid, gblock, 0, stream, mesh, fr, rl);
            else            hipLaunchKernelGGL((k_tri_spans_grouped<false, 16>), ggrid, gblock, 0, stream, mesh, fr, rl);
        }
        return;
    }
    if (rl.compact) hipLaunchKernelGGL((k_tri_spans<true>), grid, block, 0, stream, mesh, fr, rl);
    else            hipLaunchKernelGGL((k_tri_spans<false>), grid, block, 0, stream, mesh, fr, rl);
}

// Returns the variant code of the instantiation it launched (hg_last_piecewise_variant; tools/census.py): kind * 100000 + (512-slot rows) * 10000 +
// windows-or-blocks per phase * 1000 + (8-byte entries) * 100 + (bounds on the high dwords) * 10 + self-span form; kind 1 k_pw_rows,
// 3 k_pw_rows_s80, 4 k_pw_patch, 5 k_pw_tile, 6 k_pw_fused, 8 k_pw_patch with global records; + 50 for a parity-tap (map) instantiation.
int launch_pw_rows(const PwMesh &mesh, const PwFrames &fr, const RowLists &rl, uint8_t *out, int16_t *map_out, int32_t *status_next, hipStream_t stream)
{
    if (fr.n_frames <= 0 || fr.max_obj_h <= 0) return 0;
    int code = 0;
    const int rg = fr.row_group == kRowGroup ? kRowGroup : 1;
    const int nx = 1 << fr.xcc_log2;                                            // XCCs of this device (partition mode), hg_create
    const int rpx = ((fr.max_obj_h + rg - 1) / rg + nx - 1) / nx;               // row groups per XCD band
    PwFrames frs = fr;
    frs.sub_groups = sub_groups_of(fr, rpx);                                    // (sub-bands: the grid is padded to whole sub-bands)
    dim3 grid((unsigned)padded_groups(rpx, frs.sub_groups) * (unsigned)nx * (unsigned)fr.n_frames);
    // bounds :1047 on the high dwords of the rounded coordinates (hg_dev.h) whenever the source window allows it; the fp64
    // compares otherwise (negative source minimum, sources beyond 2^20 pixels a side) and in the parity-tap instantiations
    const bool hib = !fr.no_hi_bounds && hi_bounds_ok(mesh.min_src_x, (int64_t)mesh.W + mesh.min_src_x, mesh.min_src_y, (int64_t)mesh.H + mesh.min_src_y);
    const dim3 block(256);
    const size_t pad = (size_t)fr.lds_pad_kb * 1024;
#define HG_ROWS(CAP, MAPF, PHV, CMP, HB, SF) do { code = 100000 + ((CAP) > kRowSpanCapFast ? 10000 : 0) + (PHV) * 1000 + ((CMP) ? 100 : 0) + ((HB) ? 10 : 0) + (int)(SF) + ((MAPF) ? 50 : 0); \
        hipLaunchKernelGGL((k_pw_rows<CAP, MAPF, PHV, CMP, HB, SF>), grid, block, pad, stream, mesh, frs, rl, out, map_out, rpx, rg, status_next); } while (0)
#define HG_ROWS_B(CAP, PHV, CMP) do { if (hib) HG_ROWS(CAP, false, PHV, CMP, true, 0); else HG_ROWS(CAP, false, 1, CMP, false, 0); } while (0)
    if (fr.self_spans) {                                     // spans evaluated by the row workgroups themselves (sparse meshes: CAP 256, no row lists, k_tri_setup in front)
        // ONE depth: 4 windows per phase (R4.10: a wash to a slight gain over 2 for every self-span set; the census of round 6 found the
        // 1- / 2-window, the 80-SGPR and the three-edges-in-flight instantiations never picked: deleted, EXPERIMENTS.md R6.6)
        if (map_out) HG_ROWS(kRowSpanCapFast, true, 1, false, false, 1);
        else if (!hib) HG_ROWS(kRowSpanCapFast, false, 1, false, false, 1);
        else HG_ROWS(kRowSpanCapFast, false, 4, false, true, 1);
        return code;
    }
    if (rl.cap > kRowSpanCapFast) {                          // very dense meshes: 512 LDS slots per row (32 KB), one row per workgroup
        if (rl.compact) { if (map_out) HG_ROWS(kRowSpanCapDense, true, 1, true, false, false); else HG_ROWS_B(kRowSpanCapDense, 1, true); }
        else            { if (map_out) HG_ROWS(kRowSpanCapDense, true, 1, false, false, false); else HG_ROWS_B(kRowSpanCapDense, 1, false); }
        return code;
    }
    if (map_out) { if (rl.compact) HG_ROWS(kRowSpanCapFast, true, 1, true, false, false); else HG_ROWS(kRowSpanCapFast, true, 1, false, false, false); return code; }
    if (rl.compact) {                                        // dense rows: 8-byte entries
        if (fr.phase >= 2) HG_ROWS_B(kRowSpanCapFast, 2, true); else HG_ROWS_B(kRowSpanCapFast, 1, true);     // (no 4-window instantiation here)
        return code;
    }
    switch (fr.phase) {
    case 4:  HG_ROWS_B(kRowSpanCapFast, 4, false); break;
    case 2:
        if (hib && fr.sgpr_cap) { code = 302010; hipLaunchKernelGGL((k_pw_rows_s80<kRowSpanCapFast, false, 2, false, true, false>), grid, block, pad, stream,
                                                   mesh, frs, rl, out, map_out, rpx, rg, status_next); }
        else HG_ROWS_B(kRowSpanCapFast, 2, false);
        break;
    default: HG_ROWS_B(kRowSpanCapFast, 1, false); break;
    }
#undef HG_ROWS_B
#undef HG_ROWS
    return code;
}

} // namespace hg
