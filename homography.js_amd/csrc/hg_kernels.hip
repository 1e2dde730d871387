// hg_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4): the per-pixel inverse-warp hot path of
// Homography.js.  Wave64; one thread = 4 consecutive output pixels (one 16-byte RGBA8 store); fp64 coordinate math
// with contraction off so that nearest-neighbour source selection is bit-identical to the reference's JS doubles.
// Citations are file:line into the reference's Homography.js (v1.8.0).  Design notes: DESIGN.md §4.
#include "hg_kernels.h"
#include <cstdlib>
#include <type_traits>

namespace hg {

// ------------------------------------------------------------------------------------------------ helpers

// Math.round for a value already known to be finite and far below 2^52 (it passed the source bounds test).
__device__ __forceinline__ int round_inbounds(double x)
{
    double r = floor(x);
    if (x - r >= 0.5) r += 1.0;
    return (int)r;
}

__device__ __forceinline__ int64_t floordiv64(int64_t n, int64_t d)   // d > 0
{
    int64_t q = n / d;
    if ((n % d) < 0) --q;
    return q;
}

// Source fetch of the pixel loops (:1005-1007 / :1049-1052): flat index ry*W + rx into the RGBA8 array; anything
// outside the array reads `undefined` in JS and is stored as 0 in the Uint8ClampedArray.
__device__ __forceinline__ uint32_t fetch_src(const uint32_t *__restrict__ img32, int64_t n_src_px, int W, int rx, int ry)
{
    const int64_t idx = (int64_t)ry * W + rx;
    return (idx >= 0 && idx < n_src_px) ? img32[idx] : 0u;
}

// Store 4 consecutive output pixels of one row (16-byte store when the row pitch allows it).
__device__ __forceinline__ void store_quad(uint32_t *__restrict__ orow, int cq, int W, bool vec_ok, const uint32_t px[4])
{
    if (vec_ok && cq + 3 < W) {
        *reinterpret_cast<uint4 *>(orow + cq) = make_uint4(px[0], px[1], px[2], px[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) if (cq + k < W) orow[cq + k] = px[k];
    }
}

// ------------------------------------------------------------------------------------------------ k_tri_setup
// Per (frame, triangle).  Replaces _calculatePiecewiseAffineTransformMatrices :785-804, the inverseAffineMatrix loop
// :1036-1038 and the per-triangle head of fillTriangle :1113-1118.
__global__ __launch_bounds__(256) void k_tri_setup(PwMesh mesh, PwFrames fr)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (t >= mesh.n_tris) return;
    const FrameDesc fd = fr.frames[f];
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

    Seg *sg = fr.segs + ft * 3;
    Seg a, b, c;
    define_seg(d[0], d[1], d[2], d[3], a);          // p0->p1
    define_seg(d[0], d[1], d[4], d[5], b);          // p0->p2
    define_seg(d[2], d[3], d[4], d[5], c);          // p1->p2
    sg[0] = a; sg[1] = b; sg[2] = c;

    TriRange tr;
    tri_rows(d[1], d[3], d[5], tr.y_min, tr.y_end);
    tr.a = 0; tr.b = 0;
    if (tr.y_end > tr.y_min && fd.obj_w > 0 && fd.obj_h > 0) {
        // Conservative cell extent of any span of this triangle relative to its row base (y - yOff) * W:
        // intersections lie between the vertex x's (+-1 for rounding).  Absurd / non-finite input or a triangle wider
        // than the whole map (TypedArray.fill wrap-around could then straddle index 0) goes to the exact map path.
        const double x0 = d[0], x1 = d[2], x2 = d[4];
        const bool finite = fabs(x0) < 1.0e9 && fabs(x1) < 1.0e9 && fabs(x2) < 1.0e9 &&
                            fabs((double)d[1]) < 1.0e9 && fabs((double)d[3]) < 1.0e9 && fabs((double)d[5]) < 1.0e9;
        bool irregular = !finite || (tr.y_end - (int64_t)tr.y_min) > (1 << 24);
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
        if (irregular) atomicOr(&fr.status[f], FRAME_IRREGULAR);
    }
    fr.trir[ft] = tr;
}

// ------------------------------------------------------------------------------------------------ per-pixel piecewise body
// One output pixel of _inversePiecewiseAffineWarp :1044-1053 given its resolved triangle id.
struct MatCache { int id; double m[6]; };

__device__ __forceinline__ uint32_t pw_pixel(int tid_raw, int x, double y, MatCache &mc, const float *__restrict__ invm,
                                             const uint32_t *__restrict__ img32, int64_t n_src_px, int W, int H,
                                             double bx0, double bx1, double by0, double by1)
{
    const int t16 = (int)(int16_t)tid_raw;          // Int16Array element conversion (ids >= 32768 wrap, Appendix A-Q9)
    if (t16 < 0) return 0u;                         // :1045
    if (t16 != mc.id) {
        const float4 lo = *reinterpret_cast<const float4 *>(invm + (size_t)t16 * kInvStride);
        const float2 hi = *reinterpret_cast<const float2 *>(invm + (size_t)t16 * kInvStride + 4);
        mc.m[0] = lo.x; mc.m[1] = lo.y; mc.m[2] = lo.z; mc.m[3] = lo.w; mc.m[4] = hi.x; mc.m[5] = hi.y;
        mc.id = t16;
    }
    const double xd = (double)x;
    const double sx = (mc.m[0] * xd) + (mc.m[2] * y) + mc.m[4];      // :1383
    const double sy = (mc.m[1] * xd) + (mc.m[3] * y) + mc.m[5];      // :1384
    if (sx >= bx0 && sx < bx1 && sy >= by0 && sy < by1)              // :1047 (unrounded; NaN fails)
        return fetch_src(img32, n_src_px, W, round_inbounds(sx), round_inbounds(sy));   // :1048-1052
    return 0u;
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
        if (threadIdx.x == 0) atomicOr(&fr.status[f], FRAME_LDS_OVERFLOW);
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

__device__ __forceinline__ uint32_t dlo(double v) { return (uint32_t)__double2loint(v); }
// a wave-uniform double moved to scalar registers (v_cmp_f64 takes it as its scalar operand): frees two VGPRs each
__device__ __forceinline__ double sgpr_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// Output pixels are written once and never re-read by these kernels: non-temporal stores (aux bit 1 = nt) keep the
// 34 MB-per-frame output stream from displacing the shared source image in L2 / Infinity Cache (measured -13 % kernel
// time on C3 versus default-policy stores).
constexpr int kStoreNT = 2;
// k_pw_rows span key: triangle id << 14 | LDS byte offset of the span's matrix record (256 records x 48 B < 2^14), so that
// one signed max picks the last writer AND carries the address of its matrix; ids are < 2^15 (pw_fast_ok)
constexpr int kKeyShift = 14, kKeyOffMask = (1 << kKeyShift) - 1;

template <int ABL, bool COMPACT>      // ABL != 0: timing experiments only (HG_EXPERIMENTS build): 32 = no slot atomics, 64 = no entry stores
__global__ __launch_bounds__(128) void k_tri_spans(PwMesh mesh, PwFrames fr, RowLists rl)
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
            const int slot = (ABL & 32) ? (int)((t * 7 + (int)y) & 31) : atomicAdd(&rowcnt[r], 1);
            if (slot < rl.cap && !(ABL & 64)) {
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

// For 8 doubles, in place: h[i] = RTN(h[i] + 0.5), then r[i] = RTN(h[i] + M), M = 1.5 * 2^52, with the fp64 rounding mode
// switched to round-toward-minus-infinity for exactly these 16 adds.  floor(h) == floor(v + 0.5 exactly) == Math.round(v)
// for every finite double (RTN never crosses an integer upward; this also gets 0.49999999999999994 right), and it appears
// as the low dword of r.  h itself serves the bounds test:  a <= v < b  <=>  a + 0.5 <= h < b + 0.5  (a, b integers).
// (h is an in/out operand so that the 8 inputs and the 8 h share registers: 32 VGPRs for the block instead of 48.)
__device__ __forceinline__ void round_x8(double h[8], double r[8])
{
    const double M = 6755399441055744.0;
    asm volatile(
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
        "v_add_f64 %0, %0, 0.5\n\t"  "v_add_f64 %1, %1, 0.5\n\t"  "v_add_f64 %2, %2, 0.5\n\t"  "v_add_f64 %3, %3, 0.5\n\t"
        "v_add_f64 %4, %4, 0.5\n\t"  "v_add_f64 %5, %5, 0.5\n\t"  "v_add_f64 %6, %6, 0.5\n\t"  "v_add_f64 %7, %7, 0.5\n\t"
        "v_add_f64 %8, %0, %16\n\t"  "v_add_f64 %9, %1, %16\n\t"  "v_add_f64 %10, %2, %16\n\t" "v_add_f64 %11, %3, %16\n\t"
        "v_add_f64 %12, %4, %16\n\t" "v_add_f64 %13, %5, %16\n\t" "v_add_f64 %14, %6, %16\n\t" "v_add_f64 %15, %7, %16\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
        : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]),
          "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
        : "s"(M));
}

// The same for four doubles (two pixels): see STEP in k_pw_rows.
__device__ __forceinline__ void round_x4(double h[4], double r[4])
{
    const double M = 6755399441055744.0;
    asm volatile(
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
        "v_add_f64 %0, %0, 0.5\n\t" "v_add_f64 %1, %1, 0.5\n\t" "v_add_f64 %2, %2, 0.5\n\t" "v_add_f64 %3, %3, 0.5\n\t"
        "v_add_f64 %4, %0, %8\n\t"  "v_add_f64 %5, %1, %8\n\t"  "v_add_f64 %6, %2, %8\n\t"  "v_add_f64 %7, %3, %8\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
        : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
        : "s"(M));
}

// The same for two doubles (the forward tile kernels): a, b become RTN(v + 0.5); ia, ib = Math.round(v) as int32, valid while |v| < 2^31.
__device__ __forceinline__ void round_x2(double &a, double &b, int &ia, int &ib)
{
    const double M = 6755399441055744.0;
    double ra, rb;
    asm volatile(
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
        "v_add_f64 %0, %0, 0.5\n\t" "v_add_f64 %1, %1, 0.5\n\t"
        "v_add_f64 %2, %0, %4\n\t"  "v_add_f64 %3, %1, %4\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
        : "+v"(a), "+v"(b), "=&v"(ra), "=&v"(rb)
        : "s"(M));
    ia = __double2loint(ra); ib = __double2loint(rb);
}

// best[k] = max(best[k], key) for the pixels k = 0..3 (at d + 64k relative to the span start) that lie inside the span,
// i.e. (unsigned)(d + 64k) < len.  Two VALU instructions per pixel: the compare writes EXEC directly (v_cmpx) and the
// max runs under it; a compare + select + max sequence (what the compiler emits) needs three.  All 64 lanes are active
// here (uniform control flow, 256-thread blocks), EXEC is restored from the saved copy after every pixel.
template <int STRIDE = 64>
__device__ __forceinline__ void span_max4(int best[4], int d, int len, int key)
{
    const int d1 = d + STRIDE, d2 = d + 2 * STRIDE, d3 = d + 3 * STRIDE;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_lt_u32_e32 vcc, %[d0], %[len]\n\t" "v_max_i32_e32 %[b0], %[b0], %[key]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_lt_u32_e32 vcc, %[d1], %[len]\n\t" "v_max_i32_e32 %[b1], %[b1], %[key]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_lt_u32_e32 vcc, %[d2], %[len]\n\t" "v_max_i32_e32 %[b2], %[b2], %[key]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_lt_u32_e32 vcc, %[d3], %[len]\n\t" "v_max_i32_e32 %[b3], %[b3], %[key]\n\t" "s_mov_b64 exec, %[sv]"
        : [b0] "+v"(best[0]), [b1] "+v"(best[1]), [b2] "+v"(best[2]), [b3] "+v"(best[3]), [sv] "=&s"(sv)
        : [d0] "v"(d), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [len] "v"(len), [key] "v"(key)
        : "vcc");
}

// same, the four pixel offsets given explicitly
__device__ __forceinline__ void span_max4d(int best[4], int d0, int d1, int d2, int d3, int len, int key)
{
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_lt_u32_e32 vcc, %[d0], %[len]\n\t" "v_max_i32_e32 %[b0], %[b0], %[key]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_lt_u32_e32 vcc, %[d1], %[len]\n\t" "v_max_i32_e32 %[b1], %[b1], %[key]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_lt_u32_e32 vcc, %[d2], %[len]\n\t" "v_max_i32_e32 %[b2], %[b2], %[key]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_lt_u32_e32 vcc, %[d3], %[len]\n\t" "v_max_i32_e32 %[b3], %[b3], %[key]\n\t" "s_mov_b64 exec, %[sv]"
        : [b0] "+v"(best[0]), [b1] "+v"(best[1]), [b2] "+v"(best[2]), [b3] "+v"(best[3]), [sv] "=&s"(sv)
        : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [len] "v"(len), [key] "v"(key)
        : "vcc");
}

template <int CAP, int ABL, bool MAP, int PH = 1, bool COMPACT = false>
__global__ __launch_bounds__(256) void k_pw_rows(PwMesh mesh, PwFrames fr, RowLists rl, uint8_t *__restrict__ out,
                                                 int16_t *__restrict__ map_out, int groups_per_xcd, int rows_per_group,
                                                 int32_t *__restrict__ status_next)
{
    // A workgroup owns rows_per_group consecutive output rows: kRowGroup = 4 when the host expects short span lists, 1
    // for dense meshes (there, rows that share source lines should run side by side in different workgroups).  1-D grid decoded so that XCD x (= block id % 8, the observed
    // dispatch order; speed only, never correctness) walks a contiguous band of rows of one frame: vertically adjacent
    // output rows share source cache lines, which then stay in that XCD's L2 instead of being fetched by up to 8 of them.
    const int bid = blockIdx.x, xcd = bid & 7, bi = bid >> 3;
    const int f = bi / groups_per_xcd;
    const int r0 = (xcd * groups_per_xcd + (bi - f * groups_per_xcd)) * rows_per_group;
    const FrameDesc fd = fr.frames[f];
    // housekeeping for the NEXT step (saves its memset): the other parity's status words are cleared here, and below every
    // workgroup zeroes the span counters of its rows once all its waves have read them
    if (bid == 0 && status_next) for (int i = threadIdx.x; i < fr.n_frames; i += 256) status_next[i] = 0;
    if (r0 >= fd.obj_h || fd.obj_w <= 0) return;

    __shared__ __align__(16) double s_m[CAP * 6];
    __shared__ int s_lo[CAP], s_hi[CAP], s_len[CAP], s_key[CAP];   // span start / end (window overlap test), length, key (KS)
    static_assert(CAP >= 64 * kRowGroup, "packed mode gives each of the 4 rows a 64-slot block");
    constexpr int KS = CAP * 48 <= (1 << kKeyShift) ? kKeyShift : kKeyShift + 1, KMASK = (1 << KS) - 1;   // id << KS | record offset
    static_assert(CAP * 48 <= (1 << KS) && KS <= 15, "record offsets must fit below the 15-bit id");

    const int W = fd.obj_w;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform: the window loop runs on the scalar unit
    const int nwin = (W + 255) >> 8;
    const int nrows = min(rows_per_group, fd.obj_h - r0);

    // ---- span counts of the group's rows.  Packed mode (every row has at most 63 spans: the common case): all four
    // lists are loaded at once, one barrier, then wave j walks row r0 + j alone -- the list-load latency is paid once per
    // four rows and a row's span scan is a single ballot.  Otherwise the rows are taken one after the other with the
    // whole LDS (up to CAP - 1 spans) and the windows of a row are dealt to the four waves.
    int32_t *cntp = rl.cnt + (size_t)f * rl.row_stride + r0;
    int cnts[kRowGroup], cmax = 0;
#pragma unroll
    for (int j = 0; j < kRowGroup; j++) { cnts[j] = j < nrows ? cntp[j] : 0; cmax = max(cmax, cnts[j]); }
    if (cmax > rl.cap || cmax > CAP - 1) {
        __syncthreads();                                    // every wave has read the counters before they are cleared
        if ((int)threadIdx.x < nrows) cntp[threadIdx.x] = 0;
        if (threadIdx.x == 0) atomicOr(&fr.status[f], FRAME_LDS_OVERFLOW);
        return;
    }
    const bool packed = rows_per_group == kRowGroup && __builtin_amdgcn_readfirstlane(cmax) <= 63;

    // Source: raw buffer of 4*W*H bytes: an offset at or beyond its end (and the 0xffffffff of rejected pixels) returns 0
    // from the hardware range check == the JS `undefined` -> 0 of :1051.
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(frame_img(mesh, f)), 0, mesh.W * mesh.H * 4, 0x00020000);
    // :1047 on h = RTN(s + 0.5):  minSrcX <= sx < W + minSrcX  <=>  minSrcX + 0.5 <= hx < W + minSrcX + 0.5, same for y.
    // All four are tested on the doubles: the rounded coordinates are only 32-bit (a source row of 300 * 2^24 must be
    // rejected, not wrapped back into the image); what the range check of the buffer load still provides is the `undefined`
    // -> 0 of flat indices that pass :1047 and yet fall outside the array (sx in [W - 0.5, W) on the last row).
    const double bx_lo = sgpr_f64((double)mesh.min_src_x + 0.5), bx_hi = sgpr_f64((double)mesh.W + (double)mesh.min_src_x + 0.5);
    const double by_lo = sgpr_f64((double)mesh.min_src_y + 0.5), by_hi = sgpr_f64((double)mesh.H + (double)mesh.min_src_y + 0.5);
    const int pitch4 = mesh.W * 4;

    const float *__restrict__ ginv = fr.inv + (size_t)f * mesh.n_tris * kInvStride;
    // row `row` of the group -> LDS slots [base, base + cnt) (+ a NaN record in slot base + nan_slot that pixels without a
    // triangle point at); threads t0, t0 + step, ... of the caller's thread set do the copying
    auto load_row = [&](int row, int cnt, int base, int nan_slot, int t0, int step) {
        const double y = (double)(r0 + row + fd.y_off);
        // (ABL & 1, timing experiment only: the entries of one of 8 fixed groups of this XCD's band in frame 0 -- always L2-resident --
        //  with this row's own count: what the prologue would cost if the lists were cache hits)
        const size_t e0 = (ABL & 1) ? ((size_t)(xcd * groups_per_xcd + ((bi - f * groups_per_xcd) & 7)) * rows_per_group + row) * rl.cap
                                    : ((size_t)f * rl.row_stride + r0 + row) * rl.cap;
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
            s_lo[base + i] = elo; s_hi[base + i] = ehi; s_len[base + i] = ehi - elo; s_key[base + i] = ((int)id << KS) | ((base + i) * 48);
            double2 *mrec = reinterpret_cast<double2 *>(s_m + (base + i) * 6);
            mrec[0] = make_double2(m0, m2 * y);              // {m0, m2*y, m4, m1, m3*y, m5}: m2*y and m3*y are the separately
            mrec[1] = make_double2(m4, m1);                  // rounded products of :1383-1384
            mrec[2] = make_double2(m3 * y, m5);
        }
        if (t0 < 3) reinterpret_cast<double2 *>(s_m + (base + nan_slot) * 6)[t0] = make_double2(NAN, NAN);
    };

    // All 256-pixel windows w0, w0 + wstep, ... of one row whose spans sit in LDS slots [base, base + cnt).
    // PH windows per phase: the gathers of PH windows are issued (each right after its window is resolved, so they overlap
    // the next window's arithmetic), then the PH x 4 stores.  On gfx9 loads and stores share one in-order counter (vmcnt): with
    // one window per phase every wait for gather data also waits for the previous window's write acknowledgements, and only 4
    // requests per wave are ever in flight.  A streaming copy in the same instruction forms (tools/calib_fetch: 4 loads + 4
    // stores per step 4.7 TB/s, 16 + 16 per step 5.7 TB/s) shows what that costs once the source comes from HBM.
    auto do_row = [&](int row, int cnt, int base, int nan_slot, int w0, int wstep) {
        // "no triangle": smaller than every real key, its low bits address the NaN record
        const int nan_key = (int)0x80000000u | ((base + nan_slot) * 48);
        const int r = r0 + row;
        const int64_t row_px = (int64_t)r * W;
        // Output row: raw buffer of 4*W bytes, so the ragged last window needs no per-pixel guard (stores past the row
        // end are dropped by the hardware range check).
        const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(out + fd.out_off + row_px * 4, 0, W * 4, 0x00020000);
        // (16-byte stores need 16-byte aligned addresses and must not straddle the row end: the range check drops a store whole)
        const bool vec_zero = ((W & 3) == 0) && (((fd.out_off + (uint64_t)row_px * 4) & 15) == 0);
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u zero4 = { 0u, 0u, 0u, 0u };
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
                unsigned long long any = (ABL & 8) ? 1ull : 0ull;
                if (ABL & 8) { best[0] = best[1] = best[2] = best[3] = (base + w % (cnt > 0 ? cnt : 1)) * 48; }     // (experiments only) no triangle search
                else for (int j = 0; j < cnt; j += 64) {
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
#pragma unroll
                    for (int kk = 0; kk < 4; kk += STEP) {
                        double h[2 * STEP], rd[2 * STEP];
#pragma unroll
                        for (int k = kk; k < kk + STEP; k++) {
                            const double2 *mrec = reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(s_m) + (best[k] & KMASK));
                            const double2 m0 = mrec[0], m1 = mrec[1], m2 = mrec[2];
                            // (ABL & 16, timing experiment only: the access pattern of a 16 px x 4 row lane patch instead of 64 px x 1 row)
                            const double xd = (ABL & 16) ? (double)(c0 + (lane & 15) + 16 * k + fd.x_off) : xd0 + (double)(k * 64);
                            // :1383-1384  (m0*x) + (m2*y) + m4.  m0*x is exact in fp64 (24-bit f32 significand times an integer
                            // below 2^24), so fma(m0, x, m2*y) == RN((m0*x) + (m2*y)) bit for bit: one instruction instead of two.
                            h[2 * (k - kk)]     = fma(m0.x, xd, m0.y) + m1.x;
                            h[2 * (k - kk) + 1] = fma(m1.y, xd, m2.x) + m2.y;
                            if (ABL & 16) h[2 * (k - kk) + 1] += (double)(lane >> 4);
                        }
                        if constexpr (STEP == 4) round_x8(h, rd); else round_x4(h, rd);
#pragma unroll
                        for (int k = kk; k < kk + STEP; k++) {
                            const int q = 2 * (k - kk);
                            const bool inb = (int)(h[q] >= bx_lo) & (int)(h[q] < bx_hi) & (int)(h[q + 1] >= by_lo) & (int)(h[q + 1] < by_hi);   // NaN fails
                            const uint32_t o = (uint32_t)(__mul24((int)dlo(rd[q + 1]), pitch4) + ((int)dlo(rd[q]) << 2));     // :1048-1049
                            const uint32_t off = inb ? o : 0xffffffffu;
                            px[p][k] = (ABL & 2) ? off : __builtin_amdgcn_raw_buffer_load_b32(src, off, 0, 0);       // outside the array -> 0
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
                if (ABL & 4) { if ((px[p][0] ^ px[p][1] ^ px[p][2] ^ px[p][3]) != 0x9e3779b9u) continue; }
                if (empty[p] && vec_zero) {                 // 1 KB of zeros: which lane writes which pixel does not matter -> one 16-byte store per lane
                    __builtin_amdgcn_raw_buffer_store_b128(zero4, dst, ((w << 8) + lane * 4) * 4, 0, kStoreNT);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) __builtin_amdgcn_raw_buffer_store_b32(px[p][k], dst, (cq + k * 64) * 4, 0, kStoreNT);
                }
            }
        }
    };

    // One window per wave iteration.  Measured alternatives: two windows in flight per wave (74 VGPRs, 6
    // waves/SIMD) and a phased variant (4 windows resolved, then 16 gathers, then 16 stores) are both ~4 % slower: with
    // 8 waves/SIMD the other waves already cover a window's memory latency, and reads + writes together run at ~5.2 TB/s.
    const int npass = packed ? 1 : nrows;
    for (int pass = 0; pass < npass; pass++) {
        const int row = packed ? wave : pass;               // everything below is wave-uniform (scalar registers)
        const int cnt = __builtin_amdgcn_readfirstlane(cnts[0] * (row == 0) + cnts[1] * (row == 1) + cnts[2] * (row == 2) + cnts[3] * (row == 3));
        const int base = packed ? wave * 64 : 0, nan_slot = packed ? 63 : CAP - 1;
        load_row(row, cnt, base, nan_slot, packed ? lane : (int)threadIdx.x, packed ? 64 : 256);
        __syncthreads();
        if (pass == 0 && (int)threadIdx.x < nrows) cntp[threadIdx.x] = 0;
        if (row < nrows) do_row(row, cnt, base, nan_slot, packed ? 0 : wave, packed ? 1 : 4);
        if (!packed) __syncthreads();                       // the next row overwrites the records
    }
}

// ------------------------------------------------------------------------------------------------ k_pw_patch (dense meshes)
// Same contract and the same row lists as k_pw_rows, for meshes whose rows carry 64..199 spans (C5: 5 000 triangles on 8K,
// ~145 per row).  There k_pw_rows runs one row per workgroup, and under the steep shear such meshes usually have (source y
// moves ~1 row per output pixel) the 64 lanes of a row-contiguous gather hit 64 different 128-byte source lines.  Here a
// workgroup owns 4 output rows and every gather instruction covers a 2-D patch, 16 pixels x 4 rows: vertically adjacent
// output pixels read horizontally adjacent source pixels, so a patch touches ~19 lines.  What makes four dense rows fit
// in LDS: the matrix records are stored once per TRIANGLE of the group (a triangle crossing all four rows appears in four
// lists), found through a small hash table while the lists are loaded; m2*y, m3*y are then formed per pixel (one more
// fp64 multiply per coordinate, rounded exactly where the reference rounds them).  The span lookup is per lane (lanes of
// a wave sit on four different rows), through per-(row, 64-pixel column) bins of span indices built in LDS.
// Limits (host picks the kernel from its estimate; a group that exceeds one flags the frame -> map path, and the context
// stops using this kernel): <= 199 spans per row, <= 208 triangles per 4-row group, obj_w <= 8192.  A bin with more than
// 8 spans is handled inside the kernel (the block tests the row's whole list).
constexpr int kPatchRows = 4, kPatchCap = 200, kPatchRecs = 208, kPatchBins = 128, kPatchBinSlots = 8, kPatchHash = 1024, kPatchTilePitch = 68;
// (sized so that six workgroups fit a CU's 160 KB of LDS: 26.9 KB each)
static_assert(kPatchHash * 4 <= kPatchRows * kPatchBins * kPatchBinSlots, "the hash table lives in the bin-slot area");
// GLOBALREC variant for very dense meshes (up to 511 spans per row: README-scale, ~23 000 triangles on 4K): no matrix
// records in LDS at all -- a pixel reads its triangle's inverse matrix (6 floats, the tap array k_tri_spans fills, L2
// resident) from global memory and widens it itself; the LDS then holds 4 x 512 spans (30.4 KB, five workgroups per CU).
constexpr int kPatchCapDense = 512;

template <bool GLOBALREC>
__global__ __launch_bounds__(256) void k_pw_patch(PwMesh mesh, PwFrames fr, RowLists rl, uint8_t *__restrict__ out,
                                                  int groups_per_xcd, int32_t *__restrict__ status_next)
{
    constexpr int CAPR = GLOBALREC ? kPatchCapDense : kPatchCap;            // spans per row
    using slot_t = typename std::conditional<GLOBALREC, uint16_t, uint8_t>::type;
    const int bid = blockIdx.x, xcd = bid & 7, bi = bid >> 3;
    const int f = bi / groups_per_xcd;
    const int r0 = (xcd * groups_per_xcd + (bi - f * groups_per_xcd)) * kPatchRows;
    const FrameDesc fd = fr.frames[f];
    if (bid == 0 && status_next) for (int i = threadIdx.x; i < fr.n_frames; i += 256) status_next[i] = 0;   // (see k_pw_rows)
    if (r0 >= fd.obj_h || fd.obj_w <= 0) return;

    __shared__ __align__(16) double s_rec[GLOBALREC ? 6 : (kPatchRecs + 1) * 6];   // {m0, m2, m4, m1, m3, m5} per triangle; last = NaN record
    __shared__ uint32_t s_lohi[kPatchRows * CAPR];                          // span cells [lo, hi) of the row, 16 bits each
    __shared__ int s_key[kPatchRows * CAPR];                                // id << 14 | byte offset of the triangle's record (GLOBALREC: the id)
    __shared__ int s_bincnt[kPatchRows * kPatchBins];
    __shared__ __align__(4) slot_t s_bin[kPatchRows * kPatchBins * kPatchBinSlots];   // span indices per (row, 64-px column)
    __shared__ uint32_t s_tile[4 * kPatchRows * kPatchTilePitch];          // per wave: 4 rows x 64 pixels (+ padding against bank conflicts)
    __shared__ int s_nrec, s_fail;
    uint32_t *s_hash = reinterpret_cast<uint32_t *>(s_bin);                 // id << 16 | (record + 1), 0 = empty; used before the bins

    const int W = fd.obj_w;
    const int nbins = (W + 63) >> 6;
    const int nrows = min(kPatchRows, fd.obj_h - r0);
    int32_t *cntp = rl.cnt + (size_t)f * rl.row_stride + r0;
    int cnts[kPatchRows], cmax = 0;
#pragma unroll
    for (int j = 0; j < kPatchRows; j++) { cnts[j] = j < nrows ? cntp[j] : 0; cmax = max(cmax, cnts[j]); }
    for (int i = threadIdx.x; i < kPatchRows * kPatchBins; i += 256) s_bincnt[i] = 0;
    if (!GLOBALREC) for (int i = threadIdx.x; i < kPatchHash; i += 256) s_hash[i] = 0u;
    if (threadIdx.x == 0) { s_nrec = 0; s_fail = (cmax > rl.cap || cmax > CAPR - 1 || nbins > kPatchBins) ? 1 : 0; }
    __syncthreads();
    if ((int)threadIdx.x < nrows) cntp[threadIdx.x] = 0;                    // every wave has read the counters: clean for the next step
    const bool bad0 = s_fail != 0;

    // ---- phase 1: span lists -> LDS; each triangle of the group gets ONE matrix record (hash on the id: the thread that
    // claims the bucket writes the record and publishes its index; the others remember the bucket and read it later)
    const float *__restrict__ ginv0 = fr.inv + (size_t)f * mesh.n_tris * kInvStride;     // this frame's inverse matrices (tap array)
    int my_bucket[(kPatchRows * CAPR + 255) / 256];
    int n_mine = 0;
    if (!bad0) for (int e = threadIdx.x; e < kPatchRows * CAPR; e += 256) {
        const int rr = e / CAPR, i = e - rr * CAPR;
        const int cnt = cnts[0] * (rr == 0) + cnts[1] * (rr == 1) + cnts[2] * (rr == 2) + cnts[3] * (rr == 3);
        int bucket = -1;
        if (i < cnt) {
            const uint2 a = static_cast<const uint2 *>(rl.ent)[((size_t)f * rl.row_stride + r0 + rr) * rl.cap + i];      // RowEnt8 (the host pairs this kernel with compact lists)
            s_lohi[e] = a.x;
            const uint32_t id = a.y;
            if (!GLOBALREC) {
                uint32_t hpos = (id * 2654435761u) >> 22;                   // 10 bits
                for (int probe = 0; probe < kPatchHash; probe++, hpos = (hpos + 1) & (kPatchHash - 1)) {
                    const uint32_t old = atomicCAS(&s_hash[hpos], 0u, (id << 16) | 0xffffu);
                    if (old == 0u) {                                        // claimed: this thread owns the triangle's record
                        const int rec = atomicAdd(&s_nrec, 1);
                        if (rec < kPatchRecs) {
                            const float4 ma = *reinterpret_cast<const float4 *>(ginv0 + (size_t)id * kInvStride);
                            const float2 mb = *reinterpret_cast<const float2 *>(ginv0 + (size_t)id * kInvStride + 4);
                            double2 *mrec = reinterpret_cast<double2 *>(s_rec + rec * 6);
                            mrec[0] = make_double2((double)ma.x, (double)ma.z);    // m0, m2
                            mrec[1] = make_double2((double)mb.x, (double)ma.y);    // m4, m1
                            mrec[2] = make_double2((double)ma.w, (double)mb.y);    // m3, m5
                            s_hash[hpos] = (id << 16) | (uint32_t)(rec + 1);
                        } else s_fail = 1;
                        bucket = (int)hpos;
                        break;
                    }
                    if ((old >> 16) == id) { bucket = (int)hpos; break; }
                }
            }
            s_key[e] = (int)id;
        }
        my_bucket[n_mine++] = bucket;
    }
    if (!GLOBALREC && threadIdx.x < 3) reinterpret_cast<double2 *>(s_rec + kPatchRecs * 6)[threadIdx.x] = make_double2(NAN, NAN);
    __syncthreads();
    // ---- phase 2: keys (id << 14 | record offset), once every record index is published
    const bool bad1 = s_fail != 0;
    n_mine = 0;
    if (!GLOBALREC && !bad1) for (int e = threadIdx.x; e < kPatchRows * CAPR; e += 256) {
        const int bucket = my_bucket[n_mine++];
        if (bucket >= 0) s_key[e] = (s_key[e] << kKeyShift) | (int)(((s_hash[bucket] & 0xffffu) - 1u) * 48u);
    }
    if (!GLOBALREC) __syncthreads();
    // ---- phase 3: the hash table is dead, its memory becomes the bins: span index -> every 64-pixel column it overlaps
    if (!bad1) for (int e = threadIdx.x; e < kPatchRows * CAPR; e += 256) {
        const int rr = e / CAPR, i = e - rr * CAPR;
        const int cnt = cnts[0] * (rr == 0) + cnts[1] * (rr == 1) + cnts[2] * (rr == 2) + cnts[3] * (rr == 3);
        if (i < cnt) {
            const uint32_t lh = s_lohi[e];
            const int lo = (int)(lh & 0xffffu), hi = (int)(lh >> 16);
            for (int b = lo >> 6; b <= (hi - 1) >> 6 && b < nbins; b++) {
                const int pos = atomicAdd(&s_bincnt[rr * kPatchBins + b], 1);          // (a count beyond the slots marks the bin as overfull)
                if (pos < kPatchBinSlots) s_bin[(rr * kPatchBins + b) * kPatchBinSlots + pos] = (slot_t)i;
            }
        }
    }
    __syncthreads();
    if (s_fail) {                                           // the host redoes the frame through the materialised map
        if (threadIdx.x == 0) atomicOr(&fr.status[f], FRAME_LDS_OVERFLOW);
        return;
    }

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rr = lane >> 4;                               // lane = (row of the group, columns (lane & 15) + 16k of the block)
    int ck[4];
#pragma unroll
    for (int k = 0; k < 4; k++) ck[k] = (lane & 15) + 16 * k;
    const double y = (double)(r0 + rr + fd.y_off);
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(frame_img(mesh, f)), 0, mesh.W * mesh.H * 4, 0x00020000);
    // output: the group's rows as one raw buffer; lanes of rows past the frame end and pixels past the row end get an
    // offset the hardware range check drops
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(out + fd.out_off + (int64_t)r0 * W * 4, 0, nrows * W * 4, 0x00020000);
    const double bx_lo = (double)mesh.min_src_x + 0.5, bx_hi = (double)mesh.W + (double)mesh.min_src_x + 0.5;
    const double by_lo = (double)mesh.min_src_y + 0.5, by_hi = (double)mesh.H + (double)mesh.min_src_y + 0.5;
    const int pitch4 = mesh.W * 4;
    const int nan_key = GLOBALREC ? -1 : ((int)0x80000000u | (kPatchRecs * 48));
    const int row_base = rr * CAPR;
    const int my_cnt = cnts[0] * (rr == 0) + cnts[1] * (rr == 1) + cnts[2] * (rr == 2) + cnts[3] * (rr == 3);
    uint32_t *tile = s_tile + wave * (kPatchRows * kPatchTilePitch);
    const float *__restrict__ ginv = fr.inv + (size_t)f * mesh.n_tris * kInvStride;     // GLOBALREC: this frame's inverse matrices

    // One 64-pixel-wide column block, all 4 rows: span lookup, coordinates, gathers issued (not waited for).
    auto resolve_gather = [&](int cw, uint32_t px[4]) {
        const int c0 = cw << 6;                             // pixel k of the lane: (c0 + ck[k], r0 + rr)
        int best[4] = { nan_key, nan_key, nan_key, nan_key };
        const int bidx = rr * kPatchBins + cw;
        const int nb = s_bincnt[bidx];
        const slot_t *bin = s_bin + bidx * kPatchBinSlots;
        if (!__any(nb > kPatchBinSlots)) {
            for (int p = 0; __any(p < nb); p++) {
                int lo = 0, len = 0, key = 0;               // len 0: no pixel passes the span test
                if (p < nb) {
                    const int e = row_base + bin[p];
                    const uint32_t lh = s_lohi[e];
                    lo = (int)(lh & 0xffffu); len = (int)(lh >> 16) - lo; key = s_key[e];
                }
                const int d = c0 - lo;
                span_max4d(best, d + ck[0], d + ck[1], d + ck[2], d + ck[3], len, key);     // larger id wins (== last writer of :852-858)
            }
        } else {                                            // more spans in one 64-pixel bin than it has slots (slivers): test the
            for (int i = 0; __any(i < my_cnt); i++) {       // row's whole list for this block -- slow, exact, rare
                int lo = 0, len = 0, key = 0;
                if (i < my_cnt) {
                    const uint32_t lh = s_lohi[row_base + i];
                    lo = (int)(lh & 0xffffu); len = (int)(lh >> 16) - lo; key = s_key[row_base + i];
                }
                const int d = c0 - lo;
                span_max4d(best, d + ck[0], d + ck[1], d + ck[2], d + ck[3], len, key);
            }
        }
        double h[8], rd[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double m0, m1, m2, m3, m4, m5;
            if (GLOBALREC) {                                // the triangle's f32 inverse matrix straight from the tap array
                const float4 a = *reinterpret_cast<const float4 *>(ginv + (size_t)max(best[k], 0) * kInvStride);
                const float2 b = *reinterpret_cast<const float2 *>(ginv + (size_t)max(best[k], 0) * kInvStride + 4);
                const double nanq = best[k] < 0 ? NAN : 0.0;     // no triangle: every coordinate becomes NaN and fails :1047
                m0 = a.x; m1 = a.y; m2 = a.z; m3 = a.w; m4 = (double)b.x + nanq; m5 = (double)b.y + nanq;
            } else {
                const double2 *mrec = reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(s_rec) + (best[k] & kKeyOffMask));
                const double2 m02 = mrec[0], m41 = mrec[1], m35 = mrec[2];
                m0 = m02.x; m2 = m02.y; m4 = m41.x; m1 = m41.y; m3 = m35.x; m5 = m35.y;
            }
            const double xd = (double)(c0 + ck[k] + fd.x_off);
            // :1383-1384  (m0*x) + (m2*y) + m4: m2*y rounded on its own, m0*x exact in fp64 (see k_pw_rows)
            h[2 * k]     = fma(m0, xd, m2 * y) + m4;
            h[2 * k + 1] = fma(m1, xd, m3 * y) + m5;
        }
        round_x8(h, rd);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool inb = (int)(h[2 * k] >= bx_lo) & (int)(h[2 * k] < bx_hi) & (int)(h[2 * k + 1] >= by_lo) & (int)(h[2 * k + 1] < by_hi);   // :1047 (NaN fails)
            const uint32_t o = (uint32_t)(__mul24((int)dlo(rd[2 * k + 1]), pitch4) + ((int)dlo(rd[2 * k]) << 2));     // :1048-1049
            px[k] = __builtin_amdgcn_raw_buffer_load_b32(src, inb ? o : 0xffffffffu, 0, 0);
        }
    };
    // 64 x 4 transpose through this wave's LDS tile (wave-synchronous: no barrier), so that each store instruction
    // writes 256 contiguous bytes of ONE row instead of four 64-byte pieces (measured: 0.62 -> 0.50 ms on C5)
    auto transpose_store = [&](int cw, const uint32_t px[4]) {
#pragma unroll
        for (int k = 0; k < 4; k++) tile[rr * kPatchTilePitch + ck[k]] = px[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int xs = (cw << 6) + lane;
#pragma unroll
        for (int k = 0; k < 4; k++) {                       // row k of the group, pixel xs: past the row / frame end -> dropped
            const uint32_t v = tile[k * kPatchTilePitch + lane];
            __builtin_amdgcn_raw_buffer_store_b32(v, dst, (xs < W && k < nrows) ? (uint32_t)(k * W + xs) * 4u : 0xffffffffu, 0, kStoreNT);
        }
        __builtin_amdgcn_wave_barrier();
    };
    // This wave's column blocks cw = wave, wave + 4, ...  (Measured and dropped: two blocks per phase -- the gathers of block A
    // in flight while block B is resolved -- 0.504 vs 0.503 ms on C5: the wait for a block's gathers is not what limits it.)
    for (int cw = wave; cw < nbins; cw += 4) {
        uint32_t px[4];
        resolve_gather(cw, px);
        transpose_store(cw, px);
    }
}

// ------------------------------------------------------------------------------------------------ materialised-map path
// Rasteriser: one workgroup per triangle, one wave per source row y, lanes stride the span's cells.
// atomicMax over raw ids on a map initialised to -1 == sequential "last writer wins" (Appendix A-Q3).
__global__ __launch_bounds__(256) void k_map_fill(PwFrames fr, int f, int T, FrameDesc fd, int32_t *__restrict__ map32)
{
    const int t = blockIdx.x;
    const TriRange tr = fr.trir[(size_t)f * T + t];
    const Seg *segs = fr.segs + ((size_t)f * T + t) * 3;
    const int64_t len = (int64_t)fd.obj_w * fd.obj_h;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t y_first = tr.y_min, y_stop = tr.y_end;
    clamp_rows(y_first, y_stop, fd.y_off, fd.obj_w, len);
    for (int64_t y = y_first + wave; y < y_stop; y += 4) {
        int64_t k, fin;
        span_cells(segs, (double)y, (double)fd.y_off, (double)fd.obj_w, len, k, fin);
        for (int64_t c = k + lane; c < fin; c += 64) atomicMax(&map32[c], t);
    }
}

__global__ void k_fill_i32(int32_t *p, size_t n, int32_t v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

__global__ void k_map_to_i16(const int32_t *__restrict__ m32, int16_t *__restrict__ m16, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) m16[i] = (int16_t)m32[i];
}

// Pixel loop :1042-1056 reading the materialised map.  Block = 64 x 4 threads = 4 rows x 256 pixels.
__global__ __launch_bounds__(256) void k_pw_from_map(PwMesh mesh, const float *__restrict__ invm, FrameDesc fd,
                                                     const int32_t *__restrict__ map32, uint8_t *__restrict__ out)
{
    const int r = blockIdx.y * 4 + threadIdx.y;
    const int cq = (blockIdx.x * 64 + threadIdx.x) << 2;
    const int W = fd.obj_w;
    if (r >= fd.obj_h || cq >= W) return;
    const int64_t row0 = (int64_t)r * W;
    const uint32_t *__restrict__ img32 = reinterpret_cast<const uint32_t *>(mesh.img);
    const int64_t n_src_px = (int64_t)mesh.W * mesh.H;
    uint32_t *__restrict__ orow = reinterpret_cast<uint32_t *>(out + fd.out_off) + row0;
    const bool vec_ok = ((W & 3) == 0) && ((fd.out_off & 15) == 0);
    const double y = (double)(r + fd.y_off);
    const double bx0 = (double)mesh.min_src_x, bx1 = (double)mesh.W + (double)mesh.min_src_x;
    const double by0 = (double)mesh.min_src_y, by1 = (double)mesh.H + (double)mesh.min_src_y;
    uint32_t px[4];
    MatCache mc; mc.id = -1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int tid = (cq + k < W) ? map32[row0 + cq + k] : -1;
        px[k] = pw_pixel(tid, cq + k + fd.x_off, y, mc, invm, img32, n_src_px, mesh.W, mesh.H, bx0, bx1, by0, by1);
    }
    store_quad(orow, cq, W, vec_ok, px);
}

// ------------------------------------------------------------------------------------------------ k_geo
// _inverseGeometricWarp pixel loop :997-1011.  Block = 64 x 4 threads = 4 rows x 256 pixels; blockIdx.z = frame.
template <int KIND>
__global__ __launch_bounds__(256) void k_geo(const FrameDesc *__restrict__ frames, const double *__restrict__ mats,
                                             const uint8_t *__restrict__ img0, int W, int H, int n_imgs, uint64_t img_stride, uint8_t *__restrict__ out)
{
    const FrameDesc fd = frames[blockIdx.z];
    const uint8_t *__restrict__ img = n_imgs > 1 ? img0 + (uint64_t)(blockIdx.z % n_imgs) * img_stride : img0;
    const int r = blockIdx.y * 4 + threadIdx.y;
    const int cq = (blockIdx.x * 64 + threadIdx.x) << 2;
    const int OW = fd.obj_w;
    if (r >= fd.obj_h || cq >= OW) return;
    const double *__restrict__ mp = mats + (size_t)blockIdx.z * 8;
    double m[8];
#pragma unroll
    for (int k = 0; k < 8; k++) m[k] = mp[k];
    const uint32_t *__restrict__ img32 = reinterpret_cast<const uint32_t *>(img);
    const int64_t n_src_px = (int64_t)W * H;
    uint32_t *__restrict__ orow = reinterpret_cast<uint32_t *>(out + fd.out_off) + (int64_t)r * OW;
    const bool vec_ok = ((OW & 3) == 0) && ((fd.out_off & 15) == 0);
    const double y = (double)(r + fd.y_off);
    const double bw = (double)W, bh = (double)H;
    uint32_t px[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double x = (double)(cq + k + fd.x_off);
        double sx, sy;
        if (KIND == 0) apply_affine(m, x, y, sx, sy); else apply_projective(m, x, y, sx, sy);     // :999
        px[k] = 0u;
        if (sx >= 0 && sx < bw && sy >= 0 && sy < bh)                                            // :1001
            px[k] = fetch_src(img32, n_src_px, W, round_inbounds(sx), round_inbounds(sy));       // :1005-1007
    }
    store_quad(orow, cq, OW, vec_ok, px);
}

// nx / d and ny / d, IEEE-754 double, for operands in the PLAIN range: finite, d != 0, magnitudes such that the hardware
// division expansion would neither pre-scale its operands (v_div_scale) nor patch the result (v_div_fixup) -- the host
// proves that per frame (geo_plain_division()).  There the expansion is: r = rcp(d); two Newton steps on r; q = n * r;
// one correction q + (n - d*q) * r.  This is that very sequence with the reciprocal computed once for both quotients, so
// the results are the same bits as `nx / d`, `ny / d` (k_selftest_division compares them over the whole plain range).
__device__ __forceinline__ void div2_plain(double nx, double ny, double d, double &qx, double &qy)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    const double q0 = nx * r, q1 = ny * r;
    qx = fma(fma(-d, q0, nx), r, q0);
    qy = fma(fma(-d, q1, ny), r, q1);
}

// Self-test of div2_plain against the compiler's IEEE division on pseudo-random operands of the plain range (exponents of d
// in [-100, 130], of n in [-210, 130] or n == 0, random signs and mantissas, plus mantissa edge patterns).
__global__ void k_selftest_division(uint64_t seed, uint64_t n_per_thread, unsigned long long *mismatches)
{
    uint64_t s = seed ^ ((uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull);
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s * 0x2545F4914F6CDD1Dull; };
    auto make = [&](int emin, int emax, bool allow_zero) {
        const uint64_t a = next(), b = next();
        if (allow_zero && (a & 63) == 0) return (a & 64) ? -0.0 : 0.0;
        uint64_t mant = b & 0xFFFFFFFFFFFFFull;
        switch ((a >> 8) & 7) {                              // edge mantissas: all zeros, all ones, single bits
        case 0: mant = 0; break;
        case 1: mant = 0xFFFFFFFFFFFFFull; break;
        case 2: mant = 1ull << ((a >> 12) % 52); break;
        case 3: mant = 0xFFFFFFFFFFFFFull ^ (1ull << ((a >> 12) % 52)); break;
        default: break;
        }
        const int e = emin + (int)((a >> 20) % (uint64_t)(emax - emin + 1));
        const uint64_t bits = ((a >> 63) << 63) | ((uint64_t)(e + 1023) << 52) | mant;
        return __longlong_as_double((long long)bits);
    };
    unsigned long long bad = 0;
    for (uint64_t i = 0; i < n_per_thread; i++) {
        const double d = make(-100, 130, false), nx = make(-210, 130, true), ny = make(-210, 130, true);
        double qx, qy;
        div2_plain(nx, ny, d, qx, qy);
        const double rx = nx / d, ry = ny / d;
        if (!(qx == rx) || !(qy == ry)) bad++;               // (== : +0 equals -0, the sign of a zero never reaches a pixel)
    }
    if (bad) atomicAdd(mismatches, bad);
}

// k_geo_fast: same loop with the k_pw_rows pixel body (requirements checked by geo_fast_ok(): source < 2^31 bytes).
//   * lane l owns pixels c0 + l + 64k of its row: a gather instruction covers 64 consecutive output pixels;
//   * row-constant terms are computed once per lane: fl(m1*y), fl(m4*y), fl(m7*y) (projective) / fl(m2*y), fl(m3*y) (affine);
//   * affine: the matrix holds f32 values, m0*x is exact in fp64, so fma(m0, x, fl(m2*y)) rounds exactly where JS does;
//     projective: the matrix is full double, every product rounds: plain mul/add and two IEEE divides per pixel (KIND 1),
//     or div2_plain when the host has shown that no pixel of the frame set leaves the plain range (KIND 3); KIND 4 = per-frame
//     choice from the flag the device-side solve wrote (k_solve_frames);
//   * Math.round + bounds :1001 via two round-toward-minus-infinity adds per coordinate (round_x8), source through a
//     range-checked buffer load (0 outside the array), stores through a per-row buffer descriptor (no tail guards).
template <int KIND, int NW>
__global__ __launch_bounds__(256) void k_geo_fast(const FrameDesc *__restrict__ frames, const double *__restrict__ mats,
                                                  const uint8_t *__restrict__ img0, int W, int H, int n_imgs, uint64_t img_stride, uint8_t *__restrict__ out,
                                                  const int32_t *__restrict__ plain)
{
    const FrameDesc fd = frames[blockIdx.z];
    const uint8_t *__restrict__ img = n_imgs > 1 ? img0 + (uint64_t)(blockIdx.z % n_imgs) * img_stride : img0;
    // one wave per row of the block.  threadIdx.y is the same in all 64 lanes of a wave, but the compiler cannot know: made scalar
    // explicitly, or the row's output descriptor counts as divergent and every buffer_store below is wrapped in a waterfall
    // loop (v_readfirstlane x 4, two 64-bit compares, exec save / restore per store: a sixth of the kernel's instructions)
    const int r = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y);
    const int lane = threadIdx.x;
    const int cb = blockIdx.x * (256 * NW);                // this wave's NW consecutive 256-pixel windows of the row
    const int OW = fd.obj_w;
    if (r >= fd.obj_h || cb >= OW) return;
    const double *__restrict__ mp = mats + (size_t)blockIdx.z * 8;
    double m[8];
#pragma unroll
    for (int k = 0; k < 8; k++) m[k] = mp[k];
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(img), 0, W * H * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(out + fd.out_off + (int64_t)r * OW * 4, 0, OW * 4, 0x00020000);
    const double y = (double)(r + fd.y_off);
    const double bx_hi = (double)W + 0.5, by_hi = (double)H + 0.5;
    const int pitch4 = W * 4;
    // row constants, once per wave: fl(m2*y), fl(m3*y) (affine) / fl(m1*y), fl(m4*y), fl(m7*y) (projective)   :1383-1384 / :1402-1403
    const double cx = (KIND == 0 || KIND == 2) ? m[2] * y : m[1] * y, cy = (KIND == 0 || KIND == 2) ? m[3] * y : m[4] * y, ad = m[7] * y;
    // KIND 4: matrices solved on the device (k_solve_frames), which also proved (or not) the plain range per frame
    const bool use_plain = KIND == 4 && __builtin_amdgcn_readfirstlane(plain[blockIdx.z]) != 0;
    // NW windows per wave, gathers of all of them issued before the first store (see k_pw_rows: loads and stores share vmcnt)
    uint32_t px[NW][4];
#pragma unroll
    for (int p = 0; p < NW; p++) {
        const int c0 = cb + p * 256;
        if (c0 >= OW) break;                               // wave-uniform
        double h[8], rd[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double x = (double)(c0 + lane + k * 64 + fd.x_off);
            if (KIND == 0) {                               // f32-valued matrix: exact product, fma == mul then add
                h[2 * k] = fma(m[0], x, cx) + m[4];
                h[2 * k + 1] = fma(m[1], x, cy) + m[5];
            } else if (KIND == 2) {                        // arbitrary doubles: keep both roundings
                h[2 * k] = ((m[0] * x) + cx) + m[4];
                h[2 * k + 1] = ((m[1] * x) + cy) + m[5];
            } else {
                const double den = ((m[6] * x) + ad) + 1.0;
                const double nx = ((m[0] * x) + cx) + m[2], ny = ((m[3] * x) + cy) + m[5];
                if (KIND == 3 || (KIND == 4 && use_plain)) div2_plain(nx, ny, den, h[2 * k], h[2 * k + 1]);     // same bits, one reciprocal (proved range)
                else { h[2 * k] = nx / den; h[2 * k + 1] = ny / den; }
            }
        }
        round_x8(h, rd);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool inb = (int)(h[2 * k] >= 0.5) & (int)(h[2 * k] < bx_hi) & (int)(h[2 * k + 1] >= 0.5) & (int)(h[2 * k + 1] < by_hi);     // :1001 (NaN fails)
            const uint32_t off = (uint32_t)(__mul24((int)dlo(rd[2 * k + 1]), pitch4) + ((int)dlo(rd[2 * k]) << 2)); // :1005
            px[p][k] = __builtin_amdgcn_raw_buffer_load_b32(src, inb ? off : 0xffffffffu, 0, 0);
        }
    }
#pragma unroll
    for (int p = 0; p < NW; p++) {
        const int c0 = cb + p * 256;
        if (c0 >= OW) break;
#pragma unroll
        for (int k = 0; k < 4; k++) __builtin_amdgcn_raw_buffer_store_b32(px[p][k], dst, (c0 + lane + k * 64) * 4, 0, kStoreNT);
    }
}

// ------------------------------------------------------------------------------------------------ forward (scatter) paths
// _geometricWarp :911-932 and _piecewiseAffineWarp :948-972 write each SOURCE pixel to its transformed position; where
// several land on one output pixel the sequential loops keep the last writer in raster order.  On the GPU:
//   pass 1  every source pixel computes its flat destination index exactly as the JS does (Math.round, `<< 2` on
//           ToInt32, typed-array stores outside the array are dropped) and atomicMax-es its raster rank into a
//           per-output-pixel winner word (-1 = never written);
//   pass 2  every output pixel copies its winner's source pixel (0 when that read is outside the source array: the
//           reference then stores `undefined` -> 0, which still overwrites earlier writers).
// Deterministic and bit-identical to the sequential result.

__device__ __forceinline__ int64_t fwd_dst_pixel(double nx, double ny, int x_off, int y_off, int obj_w, int64_t n_dst_px)
{
    nx = js_round(nx - (double)x_off);                                  // :924 / :962
    ny = js_round(ny - (double)y_off);
    const double dst_row = (double)((int64_t)obj_w << 2);
    const int32_t sh = (int32_t)((uint32_t)js_to_int32(nx) << 2);      // `newX << 2`
    const double idx = (ny * dst_row) + (double)sh;                    // :926 / :964
    if (!(idx >= 0.0) || !(idx + 3.0 < (double)(n_dst_px * 4))) return -1;
    return (int64_t)idx >> 2;                                           // idx is a multiple of 4 whenever it is finite
}

template <int KIND>
__global__ __launch_bounds__(256) void k_fwd_scatter_geo(const double *__restrict__ mat, int W, int H, FrameDesc fd, int32_t *__restrict__ win)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    double m[8];
#pragma unroll
    for (int k = 0; k < 8; k++) m[k] = mat[k];
    double nx, ny;
    if (KIND == 0) apply_affine(m, (double)x, (double)y, nx, ny); else apply_projective(m, (double)x, (double)y, nx, ny);   // :923
    const int64_t n_dst = (int64_t)fd.obj_w * fd.obj_h;
    const int64_t p = fwd_dst_pixel(nx, ny, fd.x_off, fd.y_off, fd.obj_w, n_dst);
    if (p >= 0) atomicMax(&win[p], y * W + x);
}

__global__ __launch_bounds__(256) void k_fwd_scatter_pw(const int32_t *__restrict__ fmap, const float *__restrict__ fwd, int min_src_x, int min_src_y,
                                                        int map_w, int map_h, FrameDesc fd, int32_t *__restrict__ win)
{
    const int mx = blockIdx.x * 256 + threadIdx.x, my = blockIdx.y;
    if (mx >= map_w) return;
    const int cell = my * map_w + mx;
    const int t16 = (int)(int16_t)fmap[cell];                           // :957 (Int16Array value)
    if (t16 <= -1) return;
    const float *mf = fwd + (size_t)t16 * 6;
    const double m[6] = { mf[0], mf[1], mf[2], mf[3], mf[4], mf[5] };
    double nx, ny;
    apply_affine(m, (double)(mx + min_src_x), (double)(my + min_src_y), nx, ny);                                            // :961
    const int64_t n_dst = (int64_t)fd.obj_w * fd.obj_h;
    const int64_t p = fwd_dst_pixel(nx, ny, fd.x_off, fd.y_off, fd.obj_w, n_dst);
    if (p >= 0) atomicMax(&win[p], cell);                               // raster rank of (x, y) in the loops :955-956
}

// win holds source linear indices (geometric: y*W + x; piecewise: cell of the source-bbox map)
__global__ __launch_bounds__(256) void k_fwd_gather(const int32_t *__restrict__ win, const uint8_t *__restrict__ img, int W, int H, int piecewise,
                                                    int min_src_x, int min_src_y, int map_w, FrameDesc fd, uint8_t *__restrict__ out)
{
    const int64_t n = (int64_t)fd.obj_w * fd.obj_h;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int w = win[i];
    uint32_t px = 0u;
    if (w >= 0) {
        int64_t sidx;
        if (piecewise) { const int my = w / map_w, mx = w - my * map_w; sidx = (int64_t)(my + min_src_y) * W + (mx + min_src_x); }   // :960
        else sidx = w;
        if (sidx >= 0 && sidx < (int64_t)W * H) px = reinterpret_cast<const uint32_t *>(img)[sidx];
    }
    reinterpret_cast<uint32_t *>(out + fd.out_off)[i] = px;
}

// ------------------------------------------------------------------------------------------------ k_fwd_tiles
// The forward warps without global atomics and without the winner buffer: one workgroup per 64 x 64 tile of OUTPUT pixels
// gathers the source pixels that land in its tile.  The candidates are enumerated conservatively -- per source row the x
// interval in which the transformed point can round into the tile (four constraints, each linear in x for a fixed row; the
// tile rectangle widened by 1/64 pixel and the interval by one pixel each side), rows bounded through the inverse map of the
// rectangle's corners -- and every candidate then runs the EXACT arithmetic of the scatter kernel (same transform order,
// Math.round, `<< 2`, array-bounds drop) to get its flat destination index; only if that index falls in the tile does its
// raster rank enter an atomicMax on the tile's 4096 winner words in LDS.  Last writer in raster order == largest rank, as in
// k_fwd_scatter_geo.  Destination x just outside the window aliases into the neighbouring row of the flat array (the
// reference does not check x): tiles near the left / right edge also enumerate those aliased rectangles (kFwdWrap columns;
// the host only takes this path when no source pixel can land further out).  Then each cell copies its winner's pixel.
// Traffic per frame: source once + output once (scatter path: + 4 x the winner buffer).
template <int KIND, bool ONE>      // ONE: a single frame whose parameters travel in the kernel arguments (no upload, no sync)
__global__ __launch_bounds__(256) void k_fwd_tiles(FwdBatch batch, const uint8_t *__restrict__ img, int W, int H, uint8_t *__restrict__ out)
{
    __shared__ int s_win[kFwdTileW * kFwdTileH];
    __shared__ int s_xa[256], s_pre[256], s_wsum[4];
    const int f = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FrameDesc fd = ONE ? batch.f0 : batch.frames[f];
    const int tx0 = blockIdx.x * kFwdTileW, ty0 = blockIdx.y * kFwdTileH;
    if (tx0 >= fd.obj_w || ty0 >= fd.obj_h) return;
    const int tx1 = min(tx0 + kFwdTileW, fd.obj_w), ty1 = min(ty0 + kFwdTileH, fd.obj_h);
    for (int i = tid; i < kFwdTileW * kFwdTileH; i += 256) s_win[i] = -1;
    double m[8], iv[9];
#pragma unroll
    for (int k = 0; k < 8; k++) m[k] = ONE ? batch.p0.m[k] : batch.params[f].m[k];
#pragma unroll
    for (int k = 0; k < 9; k++) iv[k] = ONE ? batch.p0.inv[k] : batch.params[f].inv[k];
    const int use_inv = ONE ? batch.p0.use_inv : batch.params[f].use_inv;
    const double eps = 1.0 / 64.0;

#pragma unroll 1
    for (int region = 0; region < 3; region++) {
        // rounded, un-aliased destination coordinates (u, v) (offsets subtracted) of this region, and where they land: (u - ush, v + vsh)
        int u0, u1, v0, v1, ush = 0, vsh = 0;
        if (region == 0) { u0 = tx0; u1 = tx1; v0 = ty0; v1 = ty1; }
        else if (region == 1) {                                        // u in [objW, objW + wrap) of the row above lands in columns [0, wrap)
            if (tx0 >= kFwdWrap) continue;
            u0 = fd.obj_w + tx0; u1 = fd.obj_w + min(tx1, kFwdWrap); v0 = ty0 - 1; v1 = ty1 - 1; ush = fd.obj_w; vsh = 1;
        } else {                                                       // u in [-wrap, 0) of the row below lands in columns [objW - wrap, objW)
            const int lo = max(tx0, fd.obj_w - kFwdWrap);
            if (lo >= tx1) continue;
            u0 = lo - fd.obj_w; u1 = tx1 - fd.obj_w; v0 = ty0 + 1; v1 = ty1 + 1; ush = -fd.obj_w; vsh = -1;
        }
        const double fx_lo = (double)u0 + fd.x_off - 0.5 - eps, fx_hi = (double)(u1 - 1) + fd.x_off + 0.5 + eps;
        const double fy_lo = (double)v0 + fd.y_off - 0.5 - eps, fy_hi = (double)(v1 - 1) + fd.y_off + 0.5 + eps;
        int ylo = 0, yhi = H - 1;
        if (use_inv) {                                                 // source rows of the rectangle's pre-image (a convex quad)
            double mn = INFINITY, mx = -INFINITY;
            bool ok = true;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const double fx = (c & 1) ? fx_hi : fx_lo, fy = (c & 2) ? fy_hi : fy_lo;
                const double t3 = iv[3] * fx, t4 = iv[4] * fy, t6 = iv[6] * fx, t7 = iv[7] * fy;
                const double Y = t3 + t4 + iv[5], Wd = t6 + t7 + iv[8];
                const double sw = fabs(t6) + fabs(t7) + fabs(iv[8]);
                if (!(Wd > 1e-9 * sw) || !(Wd > 1e-300)) ok = false;   // at or beyond the horizon of the inverse map (or cancelled to noise)
                const double sy = Y / Wd;
                // evaluation error of sy (products and sums rounded once each): must stay well inside the 2-row padding
                if (!(4.0e-16 * (fabs(t3) + fabs(t4) + fabs(iv[5]) + fabs(sy) * sw) < 0.5 * Wd)) ok = false;
                mn = fmin(mn, sy); mx = fmax(mx, sy);
            }
            if (ok && mn == mn && mx == mx) {
                mn = floor(mn) - 2.0; mx = ceil(mx) + 2.0;
                if (mn > (double)(H - 1) || mx < 0.0) continue;
                ylo = mn < 0.0 ? 0 : (int)mn;
                yhi = mx > (double)(H - 1) ? H - 1 : (int)mx;
            }
        }
        // the four constraints are a_k x + b_k(y) >= 0 with a_k the same for every row: one reciprocal each (the +-1 pixel
        // padding of the interval absorbs its last-bit difference from a division)
        double a[4], ra[4];
        if (KIND == 0) { a[0] = m[0]; a[1] = -m[0]; a[2] = m[1]; a[3] = -m[1]; }
        else { a[0] = m[0] - fx_lo * m[6]; a[1] = fx_hi * m[6] - m[0]; a[2] = m[3] - fy_lo * m[6]; a[3] = fy_hi * m[6] - m[3]; }
#pragma unroll
        for (int k = 0; k < 4; k++) ra[k] = fabs(a[k]) < 1e-9 ? 0.0 : -1.0 / a[k];
#pragma unroll 1
        for (int ybase = ylo; ybase <= yhi; ybase += 256) {
            // (A) one lane per source row: the x interval that can reach the rectangle
            int xa = 0, len = 0;
            const int y = ybase + tid;
            if (y <= yhi) {
                const double yd = (double)y;
                double xlo = 0.0, xhi = (double)(W - 1);
                bool empty = false;
                double b[4];
                if (KIND == 0) {
                    const double cx = m[2] * yd + m[4], cy = m[3] * yd + m[5];
                    b[0] = cx - fx_lo; b[1] = fx_hi - cx; b[2] = cy - fy_lo; b[3] = fy_hi - cy;
                } else {                                               // num - L * den >= 0 (den > 0 on the whole source: host-checked)
                    b[0] = (m[1] - fx_lo * m[7]) * yd + (m[2] - fx_lo);
                    b[1] = (fx_hi * m[7] - m[1]) * yd + (fx_hi - m[2]);
                    b[2] = (m[4] - fy_lo * m[7]) * yd + (m[5] - fy_lo);
                    b[3] = (fy_hi * m[7] - m[4]) * yd + (fy_hi - m[5]);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (ra[k] == 0.0) { if (b[k] + fabs(a[k]) * (double)W + 1e-9 < 0.0) empty = true; }   // |a| < 1e-9: no x of the row can satisfy it
                    else {
                        const double x0 = b[k] * ra[k];
                        if (a[k] > 0.0) xlo = fmax(xlo, floor(x0) - 1.0); else xhi = fmin(xhi, ceil(x0) + 1.0);
                    }
                }
                if (!empty && xlo <= xhi) { xa = (int)xlo; len = (int)xhi - xa + 1; }
            }
            // inclusive prefix sum of the interval lengths over the 256 rows of this pass
            int incl = len;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
            __syncthreads();                                           // (previous pass done with s_xa / s_pre; s_win initialised)
            if (lane == 63) s_wsum[wave] = incl;
            __syncthreads();
            int base = 0;
#pragma unroll
            for (int w = 0; w < 3; w++) if (w < wave) base += s_wsum[w];
            s_xa[tid] = xa; s_pre[tid] = base + incl;
            __syncthreads();
            const int total = s_pre[255];
            // (B) candidates laid out back to back over the 256 lanes; each lane walks its row pointer forward
            int r = 0;
#pragma unroll 1
            for (int k = tid; k < total; k += 256) {
                while (s_pre[r] <= k) r++;
                const int x = s_xa[r] + (k - (r ? s_pre[r - 1] : 0));
                const double yd = (double)(ybase + r);
                double nx, ny;
                if (KIND == 0) apply_affine(m, (double)x, yd, nx, ny); else apply_projective(m, (double)x, yd, nx, ny);      // :923
                // :924-926 in the admissible range (|rounded coordinate| < 2^24: `<< 2` and the flat index are exact integers):
                // flat = v * objW + u, i.e. cell (u - ush, v + vsh) for this region's aliasing
                double uh = nx - (double)fd.x_off, vh = ny - (double)fd.y_off;                    // :924: Math.round(newX - xOffset) ...
                int ui, vi;
                round_x2(uh, vh, ui, vi);                                                         // ... exactly, through two round-down adds each
                if (!(fabs(uh) < 16777216.0 && fabs(vh) < 16777216.0)) continue;
                const int col = ui - ush, row = vi + vsh;
                if (row >= ty0 && row < ty1 && col >= tx0 && col < tx1) atomicMax(&s_win[(row - ty0) * kFwdTileW + (col - tx0)], (ybase + r) * W + x);
            }
        }
    }
    __syncthreads();
    const uint32_t *__restrict__ img32 = reinterpret_cast<const uint32_t *>(img);
    uint32_t *__restrict__ o32 = reinterpret_cast<uint32_t *>(out + fd.out_off);
    const int cx = tid & (kFwdTileW - 1);
    for (int cy = tid >> 6; cy < ty1 - ty0; cy += 4) {
        if (tx0 + cx >= tx1) continue;
        const int w = s_win[cy * kFwdTileW + cx];
        o32[(size_t)(ty0 + cy) * fd.obj_w + tx0 + cx] = w >= 0 ? img32[w] : 0u;
    }
}

// ------------------------------------------------------------------------------------------------ forward piecewise, tile-binned
// inclusive prefix sum over the 256 threads of a workgroup (two barriers; s_wsum: 4 ints of LDS)
__device__ __forceinline__ int block_scan_incl(int v, int *s_wsum, int lane, int wave)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d); if (lane >= d) v += o; }
    __syncthreads();
    if (lane == 63) s_wsum[wave] = v;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 3; w++) if (w < wave) v += s_wsum[w];
    return v;
}

__global__ void k_bbox_init(int32_t *bbox, int T)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) { bbox[4 * t] = 0x7fffffff; bbox[4 * t + 1] = 0x7fffffff; bbox[4 * t + 2] = -1; bbox[4 * t + 3] = -1; }
}

// bbox[t] = {min mx, min my, max mx, max my} over the cells of the forward triangle map whose Int16 value (the matrix index the
// pixel loop :957-961 uses) is t.  Taken from the map itself, so every quirk of its rasterisation is included by construction.
__global__ __launch_bounds__(256) void k_fmap_bbox(const int32_t *__restrict__ fmap, int map_w, int map_h, int32_t *__restrict__ bbox, int T)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 32, y = blockIdx.y;
    if (x0 >= map_w) return;
    const int x1 = min(x0 + 32, map_w);
    const int32_t *row = fmap + (size_t)y * map_w;
    int cur = -1, xs = x0;
    for (int x = x0; x <= x1; x++) {
        const int t = x < x1 ? (int)(int16_t)row[x] : -2;
        if (t != cur) {
            if (cur >= 0 && cur < T) {
                atomicMin(&bbox[4 * cur], xs); atomicMin(&bbox[4 * cur + 1], y);
                atomicMax(&bbox[4 * cur + 2], x - 1); atomicMax(&bbox[4 * cur + 3], y);
            }
            cur = t; xs = x;
        }
    }
}

// Per (t, map row) extent of t's cells: what keeps a tile's candidates close to the pixels that really use t's matrix (a
// triangle fills half of its bbox; neighbouring triangles' bboxes overlap).  Same traversal as k_fmap_bbox.
__global__ void k_rowext_init(int32_t *rowext, size_t total_rows)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total_rows) { rowext[2 * i] = 0x7fffffff; rowext[2 * i + 1] = -1; }
}

__global__ __launch_bounds__(256) void k_fmap_rowext(const int32_t *__restrict__ fmap, int map_w, int map_h, const int32_t *__restrict__ bbox,
                                                     const uint32_t *__restrict__ rowoff, int32_t *__restrict__ rowext, int T)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 32, y = blockIdx.y;
    if (x0 >= map_w) return;
    const int x1 = min(x0 + 32, map_w);
    const int32_t *row = fmap + (size_t)y * map_w;
    int cur = -1, xs = x0;
    for (int x = x0; x <= x1; x++) {
        const int t = x < x1 ? (int)(int16_t)row[x] : -2;
        if (t != cur) {
            if (cur >= 0 && cur < T) {
                const size_t i = (size_t)rowoff[cur] + (size_t)(y - bbox[4 * cur + 1]);
                atomicMin(&rowext[2 * i], xs); atomicMax(&rowext[2 * i + 1], x - 1);
            }
            cur = t; xs = x;
        }
    }
}

// One wave per (frame, triangle): bound where the triangle's pixels can land (affine image of its cell bbox, +-1 and
// rounding), split by aliasing shift k (a pixel (u, v) lands on row v + k, column u - k objW of the flat array when
// k objW <= u < (k + 1) objW), and file (t, k) under every 64 x 64 output tile that overlaps.  Anything this cannot bound
// tightly (non-finite or huge matrices / coordinates, |k| > 2, absurd widths) flags the FRAME: the host redoes it through
// the scatter path at hg_sync.
__global__ __launch_bounds__(256) void k_fwd_pw_bins(FwdPwTiles p)
{
    const int f = blockIdx.y, t = blockIdx.x * 4 + threadIdx.y, lane = threadIdx.x;
    if (t >= p.T) return;
    const int cx0 = p.bbox[4 * t], cy0 = p.bbox[4 * t + 1], cx1 = p.bbox[4 * t + 2], cy1 = p.bbox[4 * t + 3];
    if (cx1 < cx0 || cy1 < cy0) return;
    const FrameDesc fd = p.frames[f];
    if (fd.obj_w <= 0 || fd.obj_h <= 0) return;
    const float *mf = p.fwd + ((size_t)f * p.T + t) * 6;
    double m[6];
    bool bad = fd.obj_w > (1 << 24);
#pragma unroll
    for (int k = 0; k < 6; k++) { m[k] = mf[k]; if (!(fabs(m[k]) <= 1.0e6)) bad = true; }
    double umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const double x = (double)(((c & 1) ? cx1 : cx0) + p.min_src_x), y = (double)(((c & 2) ? cy1 : cy0) + p.min_src_y);
        double fx, fy;
        apply_affine(m, x, y, fx, fy);
        if (!(fabs(fx) < 1.0e7 && fabs(fy) < 1.0e7)) bad = true;
        fx -= (double)fd.x_off; fy -= (double)fd.y_off;
        umin = fmin(umin, fx); umax = fmax(umax, fx); vmin = fmin(vmin, fy); vmax = fmax(vmax, fy);
    }
    if (bad) { if (lane == 0) atomicOr(&p.status[f], FWD_FALLBACK); return; }
    const int64_t ua = (int64_t)floor(umin) - 2, ub = (int64_t)ceil(umax) + 2, va = (int64_t)floor(vmin) - 2, vb = (int64_t)ceil(vmax) + 2;
    const int64_t kmin = floordiv64(ua, fd.obj_w), kmax = floordiv64(ub, fd.obj_w);
    if (kmin < -2 || kmax > 2) { if (lane == 0) atomicOr(&p.status[f], FWD_FALLBACK); return; }
    for (int64_t k = kmin; k <= kmax; k++) {
        const int64_t c0 = (ua > k * fd.obj_w ? ua : k * fd.obj_w) - k * fd.obj_w;
        const int64_t c1 = (ub < (k + 1) * fd.obj_w - 1 ? ub : (k + 1) * fd.obj_w - 1) - k * fd.obj_w;
        const int64_t r0 = va + k > 0 ? va + k : 0, r1 = vb + k < fd.obj_h - 1 ? vb + k : fd.obj_h - 1;
        if (c0 > c1 || r0 > r1) continue;
        const int tx0 = (int)(c0 / kFwdTileW), tx1 = (int)(c1 / kFwdTileW), ty0 = (int)(r0 / kFwdTileH), ty1 = (int)(r1 / kFwdTileH);
        const int tw = tx1 - tx0 + 1, nt = tw * (ty1 - ty0 + 1);
        for (int i = lane; i < nt; i += 64) {
            const int ty = ty0 + i / tw, tx = tx0 + i % tw;
            const size_t idx = ((size_t)f * p.tsy + ty) * p.tsx + tx;
            const int slot = atomicAdd(&p.tile_cnt[idx], 1);
            if (slot < p.cap) p.tile_ent[idx * p.cap + slot] = t | (((int)k + 2) << 16);
            else atomicOr(&p.status[f], FWD_OVERFLOW);
        }
    }
}

// One workgroup per 64 x 64 output tile: same three steps as k_fwd_tiles -- conservative candidates, the reference's exact
// arithmetic per candidate (:957-964), atomicMax of the raster rank on 4096 winner words in LDS -- with the candidates coming
// from the tile's (triangle, k) entries: (0) one lane per entry: the source rows of the tile rectangle's pre-image under that
// triangle's matrix, cut to the triangle's cell bbox; (1) one lane per (entry, row): the x interval; (2) one lane per source
// pixel, kept only if the forward map assigns it to this triangle.  Both levels are laid out back to back with prefix sums, so
// twenty small triangles cost what one large one costs.
__global__ __launch_bounds__(256) void k_fwd_pw_tiles(FwdPwTiles p, const uint8_t *__restrict__ img, int W, int H, uint8_t *__restrict__ out)
{
    __shared__ int s_win[kFwdTileW * kFwdTileH];
    __shared__ double s_m[kFwdPwCapMax][6];
    __shared__ int s_et[kFwdPwCapMax], s_ek[kFwdPwCapMax], s_eylo[kFwdPwCapMax], s_epre[kFwdPwCapMax];
    __shared__ long long s_ebase[kFwdPwCapMax];                        // index of the entry's row-extent record for source row 0
    __shared__ int s_pe[256], s_py[256], s_pxa[256], s_ppre[256];
    __shared__ int s_wsum[4];
    const int f = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FrameDesc fd = p.frames[f];
    const int tx0 = blockIdx.x * kFwdTileW, ty0 = blockIdx.y * kFwdTileH;
    if (tx0 >= fd.obj_w || ty0 >= fd.obj_h) return;
    if (p.status[f] != 0) return;                                      // flagged by k_fwd_pw_bins: the host redoes this frame
    const int tx1 = min(tx0 + kFwdTileW, fd.obj_w), ty1 = min(ty0 + kFwdTileH, fd.obj_h);
    for (int i = tid; i < kFwdTileW * kFwdTileH; i += 256) s_win[i] = -1;
    const size_t tidx = ((size_t)f * p.tsy + blockIdx.y) * p.tsx + blockIdx.x;
    const int E = min(p.tile_cnt[tidx], p.cap);
    const int32_t *__restrict__ ents = p.tile_ent + tidx * p.cap;
    const float *__restrict__ fwd = p.fwd + (size_t)f * p.T * 6;
    const double eps = 1.0 / 64.0;

    // (0) one lane per entry
    int nrows = 0;
    if (tid < E) {
        const int e = ents[tid], t = e & 0xffff, k = (e >> 16) - 2;
        double m[6];
#pragma unroll
        for (int j = 0; j < 6; j++) { m[j] = fwd[(size_t)t * 6 + j]; s_m[tid][j] = m[j]; }
        const int cx0 = p.bbox[4 * t], cy0 = p.bbox[4 * t + 1], cx1 = p.bbox[4 * t + 2], cy1 = p.bbox[4 * t + 3];
        int ylo = cy0 + p.min_src_y, yhi = cy1 + p.min_src_y;
        const double fx_lo = (double)(tx0 + k * fd.obj_w) + fd.x_off - 0.5 - eps, fx_hi = (double)(tx1 - 1 + k * fd.obj_w) + fd.x_off + 0.5 + eps;
        const double fy_lo = (double)(ty0 - k) + fd.y_off - 0.5 - eps, fy_hi = (double)(ty1 - 1 - k) + fd.y_off + 0.5 + eps;
        const double det = m[0] * m[3] - m[2] * m[1];
        if (fabs(det) > 1e-300 && fabs(det) < INFINITY) {
            double mn = INFINITY, mx = -INFINITY;
            bool ok = true;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const double dx = ((c & 1) ? fx_hi : fx_lo) - m[4], dy = ((c & 2) ? fy_hi : fy_lo) - m[5];
                const double a = m[0] * dy, b = m[1] * dx;
                const double sy = (a - b) / det;                      // y of the pre-image: (m0 dy - m1 dx) / det
                if (!(8.0e-16 * (fabs(a) + fabs(b) + fabs(det) * fabs(sy)) < 0.5 * fabs(det))) ok = false;     // evaluation error vs the 2-row padding
                mn = fmin(mn, sy); mx = fmax(mx, sy);
            }
            if (ok && mn == mn && mx == mx) {
                mn = floor(mn) - 2.0; mx = ceil(mx) + 2.0;
                if (mn > (double)ylo) ylo = mn > (double)yhi ? yhi + 1 : (int)mn;
                if (mx < (double)yhi) yhi = mx < (double)ylo ? ylo - 1 : (int)mx;
            }
        }
        nrows = yhi >= ylo ? yhi - ylo + 1 : 0;
        s_et[tid] = t; s_ek[tid] = k; s_eylo[tid] = ylo;
        s_ebase[tid] = (long long)p.rowoff[t] - (long long)(cy0 + p.min_src_y);
        (void)cx0; (void)cx1;
    }
    {
        const int incl = block_scan_incl(nrows, s_wsum, lane, wave);
        s_epre[tid] = incl;
    }
    __syncthreads();
    const int R = s_epre[kFwdPwCapMax - 1];

#pragma unroll 1
    for (int q0 = 0; q0 < R; q0 += 256) {
        // (1) one lane per (entry, source row)
        int len = 0, xa = 0, pe = 0, py = 0;
        const int q = q0 + tid;
        if (q < R) {
            int lo = 0, hi = E - 1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_epre[mid] > q) hi = mid; else lo = mid + 1; }
            pe = lo;
            py = s_eylo[pe] + (q - (pe ? s_epre[pe - 1] : 0));
            const int k = s_ek[pe];
            const double m0 = s_m[pe][0], m1 = s_m[pe][1], m2 = s_m[pe][2], m3 = s_m[pe][3], m4 = s_m[pe][4], m5 = s_m[pe][5];
            const double fx_lo = (double)(tx0 + k * fd.obj_w) + fd.x_off - 0.5 - eps, fx_hi = (double)(tx1 - 1 + k * fd.obj_w) + fd.x_off + 0.5 + eps;
            const double fy_lo = (double)(ty0 - k) + fd.y_off - 0.5 - eps, fy_hi = (double)(ty1 - 1 - k) + fd.y_off + 0.5 + eps;
            const double yd = (double)py, cx = m2 * yd + m4, cy = m3 * yd + m5;
            const double a[4] = { m0, -m0, m1, -m1 }, b[4] = { cx - fx_lo, fx_hi - cx, cy - fy_lo, fy_hi - cy };
            const int2 ext = *reinterpret_cast<const int2 *>(p.rowext + 2 * (s_ebase[pe] + py));     // this triangle's cells in map row py
            double xlo = (double)(ext.x + p.min_src_x), xhi = (double)(ext.y + p.min_src_x);
            bool empty = ext.y < ext.x;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (fabs(a[j]) < 1e-9) { if (b[j] + fabs(a[j]) * fmax(fabs(xlo), fabs(xhi)) + 1e-9 < 0.0) empty = true; }   // no x of the row can satisfy it
                else {
                    const double x0 = -b[j] / a[j];
                    if (a[j] > 0.0) xlo = fmax(xlo, floor(x0) - 1.0); else xhi = fmin(xhi, ceil(x0) + 1.0);
                }
            }
            if (!empty && xlo <= xhi) { xa = (int)xlo; len = (int)xhi - xa + 1; }
        }
        const int incl = block_scan_incl(len, s_wsum, lane, wave);      // (its first barrier also fences the previous pass's readers)
        s_pe[tid] = pe; s_py[tid] = py; s_pxa[tid] = xa; s_ppre[tid] = incl;
        __syncthreads();
        const int C = s_ppre[255];
        // (2) one lane per candidate source pixel
        int r = 0;
#pragma unroll 1
        for (int c = tid; c < C; c += 256) {
            while (s_ppre[r] <= c) r++;
            const int x = s_pxa[r] + (c - (r ? s_ppre[r - 1] : 0)), y = s_py[r], e = s_pe[r];
            const int t = s_et[e], k = s_ek[e];
            const int64_t cell = (int64_t)(y - p.min_src_y) * p.map_w + (x - p.min_src_x);
            if ((int)(int16_t)p.fmap[cell] != t) continue;                                  // :957: this pixel uses another triangle's matrix
            const double m[6] = { s_m[e][0], s_m[e][1], s_m[e][2], s_m[e][3], s_m[e][4], s_m[e][5] };
            double nx, ny;
            apply_affine(m, (double)x, (double)y, nx, ny);                                  // :961
            double uh = nx - (double)fd.x_off, vh = ny - (double)fd.y_off;                               // :962
            int ui, vi;
            round_x2(uh, vh, ui, vi);
            if (!(fabs(uh) < 1.0e9 && fabs(vh) < 1.0e9)) continue;
            const int col = ui - k * fd.obj_w, row = vi + k;
            if (row >= ty0 && row < ty1 && col >= tx0 && col < tx1) atomicMax(&s_win[(row - ty0) * kFwdTileW + (col - tx0)], (int)cell);
        }
    }
    __syncthreads();
    const uint32_t *__restrict__ img32 = reinterpret_cast<const uint32_t *>(img);
    uint32_t *__restrict__ o32 = reinterpret_cast<uint32_t *>(out + fd.out_off);
    const int cxl = tid & (kFwdTileW - 1);
    for (int cyl = tid >> 6; cyl < ty1 - ty0; cyl += 4) {
        if (tx0 + cxl >= tx1) continue;
        const int w = s_win[cyl * kFwdTileW + cxl];
        uint32_t px = 0u;
        if (w >= 0) {
            const int my = w / p.map_w, mx = w - my * p.map_w;
            const int64_t sidx = (int64_t)(my + p.min_src_y) * W + (mx + p.min_src_x);                    // :960
            if (sidx >= 0 && sidx < (int64_t)W * H) px = img32[sidx];
        }
        o32[(size_t)(ty0 + cyl) * fd.obj_w + tx0 + cxl] = px;
    }
}

// ------------------------------------------------------------------------------------------------ launchers

void launch_tri_setup(const PwMesh &mesh, const PwFrames &fr, hipStream_t stream)
{
    if (mesh.n_tris <= 0 || fr.n_frames <= 0) return;
    dim3 grid((mesh.n_tris + 255) / 256, fr.n_frames);
    hipLaunchKernelGGL(k_tri_setup, grid, dim3(256), 0, stream, mesh, fr);
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
    const dim3 grid(mesh.n_tris, fr.n_frames), block(fr.tri_threads == 64 ? 64 : 128);
#ifdef HG_EXPERIMENTS
    static const int abl = getenv("HG_ABLATE_TRI") ? atoi(getenv("HG_ABLATE_TRI")) : 0;
    if (abl == 32) { hipLaunchKernelGGL((k_tri_spans<32, false>), grid, block, 0, stream, mesh, fr, rl); return; }
    if (abl == 64) { hipLaunchKernelGGL((k_tri_spans<64, false>), grid, block, 0, stream, mesh, fr, rl); return; }
    if (abl == 96) { hipLaunchKernelGGL((k_tri_spans<96, false>), grid, block, 0, stream, mesh, fr, rl); return; }
#endif
    if (rl.compact) hipLaunchKernelGGL((k_tri_spans<0, true>), grid, block, 0, stream, mesh, fr, rl);
    else            hipLaunchKernelGGL((k_tri_spans<0, false>), grid, block, 0, stream, mesh, fr, rl);
}

void launch_pw_patch(const PwMesh &mesh, const PwFrames &fr, const RowLists &rl, uint8_t *out, int32_t *status_next, bool global_records, hipStream_t stream)
{
    if (fr.n_frames <= 0 || fr.max_obj_h <= 0) return;
    const int gpx = ((fr.max_obj_h + kPatchRows - 1) / kPatchRows + 7) / 8;
    const dim3 grid((unsigned)gpx * 8u * (unsigned)fr.n_frames);
    if (global_records) hipLaunchKernelGGL(k_pw_patch<true>, grid, dim3(256), 0, stream, mesh, fr, rl, out, gpx, status_next);
    else                hipLaunchKernelGGL(k_pw_patch<false>, grid, dim3(256), 0, stream, mesh, fr, rl, out, gpx, status_next);
}

void launch_pw_rows(const PwMesh &mesh, const PwFrames &fr, const RowLists &rl, uint8_t *out, int16_t *map_out, int32_t *status_next, hipStream_t stream)
{
    if (fr.n_frames <= 0 || fr.max_obj_h <= 0) return;
    const int rg = fr.row_group == kRowGroup ? kRowGroup : 1;
    const int rpx = ((fr.max_obj_h + rg - 1) / rg + 7) / 8;                     // row groups per XCD band
    dim3 grid((unsigned)rpx * 8u * (unsigned)fr.n_frames);
#define HG_ROWS(CAP, MAPF, PHV, CMP) hipLaunchKernelGGL((k_pw_rows<CAP, 0, MAPF, PHV, CMP>), grid, dim3(256), 0, stream, mesh, fr, rl, out, map_out, rpx, rg, status_next)
    if (rl.cap > kRowSpanCapFast) {                          // very dense meshes: 512 LDS slots per row (32 KB), one row per workgroup
        if (rl.compact) { if (map_out) HG_ROWS(kRowSpanCapDense, true, 1, true); else HG_ROWS(kRowSpanCapDense, false, 1, true); }
        else            { if (map_out) HG_ROWS(kRowSpanCapDense, true, 1, false); else HG_ROWS(kRowSpanCapDense, false, 1, false); }
        return;
    }
    if (map_out) { if (rl.compact) HG_ROWS(kRowSpanCapFast, true, 1, true); else HG_ROWS(kRowSpanCapFast, true, 1, false); return; }
    if (rl.compact) {                                        // dense rows: 8-byte entries
        if (fr.phase >= 2) HG_ROWS(kRowSpanCapFast, false, 2, true); else HG_ROWS(kRowSpanCapFast, false, 1, true);     // (no 4-window instantiation here)
        return;
    }
#ifdef HG_EXPERIMENTS
    // Timing experiments of DESIGN.md §6 (ablated variants produce WRONG pixels): only in the separate experiments build
    // (`make experiments` -> lib/libhgwarp_exp.so); the shipped library has neither the instantiations nor the switch.
    static const int abl = getenv("HG_ABLATE") ? atoi(getenv("HG_ABLATE")) : 0;
    switch (abl) {
    case 1: hipLaunchKernelGGL((k_pw_rows<kRowSpanCapFast, 1, false>), grid, dim3(256), 0, stream, mesh, fr, rl, out, map_out, rpx, rg, status_next); return;
    case 2: hipLaunchKernelGGL((k_pw_rows<kRowSpanCapFast, 2, false>), grid, dim3(256), 0, stream, mesh, fr, rl, out, map_out, rpx, rg, status_next); return;
    case 4: hipLaunchKernelGGL((k_pw_rows<kRowSpanCapFast, 4, false>), grid, dim3(256), 0, stream, mesh, fr, rl, out, map_out, rpx, rg, status_next); return;
    case 6: hipLaunchKernelGGL((k_pw_rows<kRowSpanCapFast, 6, false>), grid, dim3(256), 0, stream, mesh, fr, rl, out, map_out, rpx, rg, status_next); return;
    case 8: hipLaunchKernelGGL((k_pw_rows<kRowSpanCapFast, 8, false>), grid, dim3(256), 0, stream, mesh, fr, rl, out, map_out, rpx, rg, status_next); return;
    case 16: hipLaunchKernelGGL((k_pw_rows<kRowSpanCapFast, 16, false>), grid, dim3(256), 0, stream, mesh, fr, rl, out, map_out, rpx, rg, status_next); return;
    case 14: hipLaunchKernelGGL((k_pw_rows<kRowSpanCapFast, 14, false>), grid, dim3(256), 0, stream, mesh, fr, rl, out, map_out, rpx, rg, status_next); return;
    default: break;
    }
#endif
    switch (fr.phase) {
    case 4:  HG_ROWS(kRowSpanCapFast, false, 4, false); break;
    case 2:  HG_ROWS(kRowSpanCapFast, false, 2, false); break;
    default: HG_ROWS(kRowSpanCapFast, false, 1, false); break;
    }
#undef HG_ROWS
}

void launch_map_build(const PwMesh &mesh, const PwFrames &fr, int f, const FrameDesc &fd, int32_t *map32, hipStream_t stream)
{
    const size_t n = (fd.obj_w > 0 && fd.obj_h > 0) ? (size_t)fd.obj_w * fd.obj_h : 0;
    if (n == 0) return;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_fill_i32, dim3(blocks), dim3(256), 0, stream, map32, n, (int32_t)-1);      // :850
    if (mesh.n_tris > 0)
        hipLaunchKernelGGL(k_map_fill, dim3(mesh.n_tris), dim3(256), 0, stream, fr, f, mesh.n_tris, fd, map32);
}

void launch_pw_from_map(const PwMesh &mesh, const PwFrames &fr, int f, const FrameDesc &fd, const int32_t *map32,
                        uint8_t *out, hipStream_t stream)
{
    if (fd.obj_w <= 0 || fd.obj_h <= 0) return;
    dim3 grid((fd.obj_w + 255) / 256, (fd.obj_h + 3) / 4);
    hipLaunchKernelGGL(k_pw_from_map, grid, dim3(64, 4), 0, stream, mesh,
                       (const float *)(fr.inv + (size_t)f * mesh.n_tris * kInvStride), fd, map32, out);
}

void launch_map_to_i16(const int32_t *map32, int16_t *map16, size_t n, hipStream_t stream)
{
    if (n == 0) return;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_map_to_i16, dim3(blocks), dim3(256), 0, stream, map32, map16, n);
}

void launch_geo(int kind, bool f32_exact, const FrameDesc *frames, const double *mats, int n_frames, int max_w, int max_h,
                const uint8_t *img, int W, int H, int n_imgs, uint64_t img_stride, uint8_t *out, const int32_t *plain, int nw, hipStream_t stream)
{
    if (n_frames <= 0 || max_w <= 0 || max_h <= 0) return;
    const bool fast = ((int64_t)H + 2) * W * 4 < ((int64_t)1 << 31) && W < (1 << 21) && H < (1 << 22) && max_w < (1 << 28);
    if (fast) {
        const int NW = nw == 1 ? 1 : (nw == 2 ? 2 : 4);        // windows per wave (measured on C2, 1 -> 2 -> 4: 0.198 -> 0.177 -> 0.171 ms)
        dim3 grid((max_w + 256 * NW - 1) / (256 * NW), (max_h + 3) / 4, n_frames);
#define HG_GEO(K) do { if (NW == 1) hipLaunchKernelGGL((k_geo_fast<K, 1>), grid, dim3(64, 4), 0, stream, frames, mats, img, W, H, n_imgs, img_stride, out, plain); \
                       else if (NW == 4) hipLaunchKernelGGL((k_geo_fast<K, 4>), grid, dim3(64, 4), 0, stream, frames, mats, img, W, H, n_imgs, img_stride, out, plain); \
                       else hipLaunchKernelGGL((k_geo_fast<K, 2>), grid, dim3(64, 4), 0, stream, frames, mats, img, W, H, n_imgs, img_stride, out, plain); } while (0)
        if (kind == 1 && plain) HG_GEO(4);
        else if (kind == 1 && f32_exact) HG_GEO(3);
        else if (kind == 1) HG_GEO(1);
        else if (f32_exact) HG_GEO(0);
        else                HG_GEO(2);
#undef HG_GEO
        return;
    }
    dim3 grid((max_w + 255) / 256, (max_h + 3) / 4, n_frames);
    if (kind == 0) hipLaunchKernelGGL(k_geo<0>, grid, dim3(64, 4), 0, stream, frames, mats, img, W, H, n_imgs, img_stride, out);
    else           hipLaunchKernelGGL(k_geo<1>, grid, dim3(64, 4), 0, stream, frames, mats, img, W, H, n_imgs, img_stride, out);
}

// ------------------------------------------------------------------------------------------------ k_solve_frames
// The solve the reference repeats at the head of every inverse warp (:994: calculateTransformMatrix(dstPoints, srcPoints)),
// one lane per frame: projective = the 8x8 DLT system through numeric.js' LU in its exact operation order
// (solve_projective_regs, hg_math.h: all in registers), affine = the closed form of affineMatrixFromTriangles (f32 result,
// widened).  Also decides per projective frame whether the shared-reciprocal division is admissible for its window.
// MFMA is not used on purpose: the order of the ~500 roundings of the LU is observable in the result.
__global__ __launch_bounds__(64) void k_solve_frames(int kind, const float *__restrict__ from, const float *__restrict__ to,
                                                     const FrameDesc *__restrict__ frames, double *__restrict__ mats, int32_t *__restrict__ plain, int n)
{
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= n) return;
    double m[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (kind == 1) {
        float s[8], d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { s[k] = from[8 * (size_t)f + k]; d[k] = to[8 * (size_t)f + k]; }
        solve_projective_regs(s, d, m);
        const FrameDesc fd = frames[f];
        plain[f] = projective_plain_range(m, fd.x_off, fd.y_off, fd.obj_w, fd.obj_h) ? 1 : 0;
    } else {
        float s[6], d[6], o[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { s[k] = from[6 * (size_t)f + k]; d[k] = to[6 * (size_t)f + k]; }
        solve_affine(s, d, o);
#pragma unroll
        for (int k = 0; k < 6; k++) m[k] = o[k];
        plain[f] = 0;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) mats[8 * (size_t)f + k] = m[k];
}

void launch_solve_frames(int kind, const float *from, const float *to, const FrameDesc *frames, double *mats, int32_t *plain, int n, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_solve_frames, dim3((n + 63) / 64), dim3(64), 0, stream, kind, from, to, frames, mats, plain, n);
}

unsigned long long run_selftest_division(uint64_t seed, uint64_t samples, unsigned long long *d_counter, hipStream_t stream)
{
    const uint64_t threads = 256ull * 1024ull, per = (samples + threads - 1) / threads;
    unsigned long long h = ~0ull;                            // (stays "all wrong" if any step fails)
    if (hipMemsetAsync(d_counter, 0, sizeof(unsigned long long), stream) != hipSuccess) return h;
    hipLaunchKernelGGL(k_selftest_division, dim3(1024), dim3(256), 0, stream, seed, per, d_counter);
    if (hipMemcpyAsync(&h, d_counter, sizeof h, hipMemcpyDeviceToHost, stream) != hipSuccess) return ~0ull;
    if (hipStreamSynchronize(stream) != hipSuccess) return ~0ull;
    return h;
}

void launch_fwd_geo(int kind, const double *d_mat, const uint8_t *img, int W, int H, const FrameDesc &fd, int32_t *win, uint8_t *out, hipStream_t stream)
{
    const int64_t n = (fd.obj_w > 0 && fd.obj_h > 0) ? (int64_t)fd.obj_w * fd.obj_h : 0;
    if (n == 0) return;
    const int fb = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_fill_i32, dim3(fb), dim3(256), 0, stream, win, (size_t)n, (int32_t)-1);
    dim3 grid((W + 255) / 256, H);
    if (kind == 0) hipLaunchKernelGGL(k_fwd_scatter_geo<0>, grid, dim3(256), 0, stream, d_mat, W, H, fd, win);
    else           hipLaunchKernelGGL(k_fwd_scatter_geo<1>, grid, dim3(256), 0, stream, d_mat, W, H, fd, win);
    hipLaunchKernelGGL(k_fwd_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, win, img, W, H, 0, 0, 0, 1, fd, out);
}

void launch_fwd_tiles(int kind, const FwdBatch &batch, int n_frames, int max_w, int max_h,
                      const uint8_t *img, int W, int H, uint8_t *out, hipStream_t stream)
{
    if (n_frames <= 0 || max_w <= 0 || max_h <= 0) return;
    const dim3 grid((max_w + kFwdTileW - 1) / kFwdTileW, (max_h + kFwdTileH - 1) / kFwdTileH, n_frames);
    if (batch.params) {
        if (kind == 0) hipLaunchKernelGGL((k_fwd_tiles<0, false>), grid, dim3(256), 0, stream, batch, img, W, H, out);
        else           hipLaunchKernelGGL((k_fwd_tiles<1, false>), grid, dim3(256), 0, stream, batch, img, W, H, out);
    } else {
        if (kind == 0) hipLaunchKernelGGL((k_fwd_tiles<0, true>), grid, dim3(256), 0, stream, batch, img, W, H, out);
        else           hipLaunchKernelGGL((k_fwd_tiles<1, true>), grid, dim3(256), 0, stream, batch, img, W, H, out);
    }
}

void launch_fmap_bbox(const int32_t *fmap, int map_w, int map_h, int32_t *bbox, int T, hipStream_t stream)
{
    if (T <= 0) return;
    hipLaunchKernelGGL(k_bbox_init, dim3((T + 255) / 256), dim3(256), 0, stream, bbox, T);
    if (map_w > 0 && map_h > 0)
        hipLaunchKernelGGL(k_fmap_bbox, dim3((map_w + 32 * 256 - 1) / (32 * 256), map_h), dim3(256), 0, stream, fmap, map_w, map_h, bbox, T);
}

void launch_fmap_rowext(const int32_t *fmap, int map_w, int map_h, const int32_t *bbox, const uint32_t *rowoff, int32_t *rowext, size_t total_rows, int T, hipStream_t stream)
{
    if (T <= 0 || total_rows == 0) return;
    hipLaunchKernelGGL(k_rowext_init, dim3((unsigned)((total_rows + 255) / 256)), dim3(256), 0, stream, rowext, total_rows);
    if (map_w > 0 && map_h > 0)
        hipLaunchKernelGGL(k_fmap_rowext, dim3((map_w + 32 * 256 - 1) / (32 * 256), map_h), dim3(256), 0, stream, fmap, map_w, map_h, bbox, rowoff, rowext, T);
}

void launch_fwd_pw_tiles(const FwdPwTiles &p, int n_frames, int max_w, int max_h, const uint8_t *img, int W, int H, uint8_t *out, hipStream_t stream)
{
    if (n_frames <= 0 || max_w <= 0 || max_h <= 0) return;
    if (p.T > 0) hipLaunchKernelGGL(k_fwd_pw_bins, dim3((p.T + 3) / 4, n_frames), dim3(64, 4), 0, stream, p);
    hipLaunchKernelGGL(k_fwd_pw_tiles, dim3(p.tsx, p.tsy, n_frames), dim3(256), 0, stream, p, img, W, H, out);
}

void launch_fwd_pw(const int32_t *fmap, const float *fwd, const uint8_t *img, int W, int H, int min_src_x, int min_src_y, int map_w, int map_h,
                   const FrameDesc &fd, int32_t *win, uint8_t *out, hipStream_t stream)
{
    const int64_t n = (fd.obj_w > 0 && fd.obj_h > 0) ? (int64_t)fd.obj_w * fd.obj_h : 0;
    if (n == 0) return;
    const int fb = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_fill_i32, dim3(fb), dim3(256), 0, stream, win, (size_t)n, (int32_t)-1);
    if (map_w > 0 && map_h > 0)
        hipLaunchKernelGGL(k_fwd_scatter_pw, dim3((map_w + 255) / 256, map_h), dim3(256), 0, stream, fmap, fwd, min_src_x, min_src_y, map_w, map_h, fd, win);
    hipLaunchKernelGGL(k_fwd_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, win, img, W, H, 1, min_src_x, min_src_y, map_w > 0 ? map_w : 1, fd, out);
}

} // namespace hg
