// hg_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4): the per-pixel inverse-warp hot path of
// Homography.js.  Wave64; one thread = 4 consecutive output pixels (one 16-byte RGBA8 store); fp64 coordinate math
// with contraction off so that nearest-neighbour source selection is bit-identical to the reference's JS doubles.
// Citations are file:line into the reference's Homography.js (v1.8.0).  Design notes: DESIGN.md §4.
#include "hg_kernels.h"

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
    const uint32_t *__restrict__ img32 = reinterpret_cast<const uint32_t *>(mesh.img);
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
    for (int64_t y = (int64_t)tr.y_min + wave; y < tr.y_end; y += 4) {
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
                                             const uint8_t *__restrict__ img, int W, int H, uint8_t *__restrict__ out)
{
    const FrameDesc fd = frames[blockIdx.z];
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

void launch_geo(int kind, const FrameDesc *frames, const double *mats, int n_frames, int max_w, int max_h,
                const uint8_t *img, int W, int H, uint8_t *out, hipStream_t stream)
{
    if (n_frames <= 0 || max_w <= 0 || max_h <= 0) return;
    dim3 grid((max_w + 255) / 256, (max_h + 3) / 4, n_frames);
    if (kind == 0) hipLaunchKernelGGL(k_geo<0>, grid, dim3(64, 4), 0, stream, frames, mats, img, W, H, out);
    else           hipLaunchKernelGGL(k_geo<1>, grid, dim3(64, 4), 0, stream, frames, mats, img, W, H, out);
}

} // namespace hg
