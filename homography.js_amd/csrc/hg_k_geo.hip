// hg_k_geo.hip -- inverse affine / projective warp: k_geo, k_geo_fast, the per-frame device-side solves (k_solve_frames)
// Hand-written HIP for gfx950 (MI355X / CDNA4), wave64; fp64 coordinate math with contraction off so that nearest-neighbour
// source selection is bit-identical to the reference's JS doubles.
// Citations are file:line into the reference's Homography.js (v1.8.0).  Design notes: DESIGN.md §4.
#include "hg_dev.h"
#include <type_traits>

namespace hg {

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
                                                  const int32_t *__restrict__ plain, int xcc_log2, int groups_per_xcd, int chunks, int rotate)
{
    // 1-D grid decoded like k_pw_rows': XCD x (= block id % number of XCCs) walks a contiguous band of 4-row groups of one frame, so
    // that vertically adjacent output rows -- which share source cache lines wherever the map is not an exact row copy -- meet in
    // ONE L2 instead of being fetched by up to 8 of them (round 3: fabric reads of C2 with one source per frame 1.39 x algorithmic before).
    const int bid = blockIdx.x, xcd = bid & ((1 << xcc_log2) - 1), bi = bid >> xcc_log2;
    const int chunk = bi % chunks, bj = bi / chunks;
    // `rotate` (one source per frame): XCD x takes band (x + frame) mod XCCs.  C2's top and bottom bands miss the source on a third of
    // their pixels -- with fixed bands the XCDs holding the middle of every frame finish last (0.220 ms against 0.195 in plain block order);
    // rotated, the bands' HBM reads stay at 1.03 x the lines the frame touches (plain order 1.40 x) and the kernel takes 0.188 ms.
    const int fz = bj / groups_per_xcd, band = (xcd + (rotate ? fz : 0)) & ((1 << xcc_log2) - 1), rgroup = band * groups_per_xcd + (bj - fz * groups_per_xcd);
    const FrameDesc fd = frames[fz];
    const uint8_t *__restrict__ img = n_imgs > 1 ? img0 + (uint64_t)(fz % n_imgs) * img_stride : img0;
    // one wave per row of the block.  threadIdx.y is the same in all 64 lanes of a wave, but the compiler cannot know: made scalar
    // explicitly, or the row's output descriptor counts as divergent and every buffer_store below is wrapped in a waterfall
    // loop (v_readfirstlane x 4, two 64-bit compares, exec save / restore per store: a sixth of the kernel's instructions)
    const int r = rgroup * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y);
    const int lane = threadIdx.x;
    const int cb = chunk * (256 * NW);                     // this wave's NW consecutive 256-pixel windows of the row
    const int OW = fd.obj_w;
    if (r >= fd.obj_h || cb >= OW) return;
    const double *__restrict__ mp = mats + (size_t)fz * 8;
    double m[8];
#pragma unroll
    for (int k = 0; k < 8; k++) m[k] = mp[k];
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(img), 4, W * H, 0x00020000);       // (records of 4 bytes: gathers by pixel index)
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(out + fd.out_off + (int64_t)r * OW * 4, 0, OW * 4, 0x00020000);
    const double y = (double)(r + fd.y_off);
    // :1001 on the high dwords of h = RTN(s + 0.5) (hg_dev.h; the launcher admits only W, H < 2^20 here)
    const HiBounds hb = make_hi_bounds(0.5, (double)W + 0.5, 0.5, (double)H + 0.5);
    // row constants, once per wave: fl(m2*y), fl(m3*y) (affine) / fl(m1*y), fl(m4*y), fl(m7*y) (projective)   :1383-1384 / :1402-1403
    const double cx = (KIND == 0 || KIND == 2) ? m[2] * y : m[1] * y, cy = (KIND == 0 || KIND == 2) ? m[3] * y : m[4] * y, ad = m[7] * y;
    // KIND 4: matrices solved on the device (k_solve_frames), which also proved (or not) the plain range per frame
    const bool use_plain = KIND == 4 && __builtin_amdgcn_readfirstlane(plain[fz]) != 0;
    // NW windows per wave, gathers of all of them issued before the first store (see k_pw_rows: loads and stores share vmcnt)
    uint32_t px[NW][4];
    // (the plain / IEEE choice of KIND 4 is wave-uniform: taken once around the whole gather loop, not once per pixel -- the per-pixel
    // form kept every pixel's dependent chain behind a scalar branch)
    auto gather = [&](auto plain_tag) {
        constexpr bool PLAIN = decltype(plain_tag)::value;
#pragma unroll
        for (int p = 0; p < NW; p++) {
            const int c0 = cb + p * 256;
            if (c0 >= OW) break;                           // wave-uniform
            double h[8], rd[8];
            const double x0 = (double)(c0 + lane + fd.x_off);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double x = x0 + (double)(k * 64);    // (exact: integers far below 2^53)
                if (KIND == 0) {                           // f32-valued matrix: exact product, fma == mul then add
                    h[2 * k] = fma(m[0], x, cx) + m[4];
                    h[2 * k + 1] = fma(m[1], x, cy) + m[5];
                } else if (KIND == 2) {                    // arbitrary doubles: keep both roundings
                    h[2 * k] = ((m[0] * x) + cx) + m[4];
                    h[2 * k + 1] = ((m[1] * x) + cy) + m[5];
                } else {
                    const double den = ((m[6] * x) + ad) + 1.0;
                    const double nx = ((m[0] * x) + cx) + m[2], ny = ((m[3] * x) + cy) + m[5];
                    if (PLAIN) div2_plain(nx, ny, den, h[2 * k], h[2 * k + 1]);     // same bits, one reciprocal (proved range)
                    else { h[2 * k] = nx / den; h[2 * k + 1] = ny / den; }
                }
            }
            round_x8(h, rd);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool inb = hi_inb(hb, h[2 * k], h[2 * k + 1]);                                     // :1001 (NaN fails)
                const int idx = __mul24((int)dlo(rd[2 * k + 1]), W) + (int)dlo(rd[2 * k]);                            // :1005, in pixels
                px[p][k] = hg_struct_load_u32(src, inb ? idx : -1, 0, 0, 0);
            }
        }
    };
    if (KIND == 3 || (KIND == 4 && use_plain)) gather(std::true_type{}); else gather(std::false_type{});
#pragma unroll
    for (int p = 0; p < NW; p++) {
        const int c0 = cb + p * 256;
        if (c0 >= OW) break;
#pragma unroll
        for (int k = 0; k < 4; k++) __builtin_amdgcn_raw_buffer_store_b32(px[p][k], dst, (c0 + lane + k * 64) * 4, 0, kStoreNT);
    }
}

void launch_geo(int kind, bool f32_exact, const FrameDesc *frames, const double *mats, int n_frames, int max_w, int max_h,
                const uint8_t *img, int W, int H, int n_imgs, uint64_t img_stride, uint8_t *out, const int32_t *plain, int nw, int xcc_log2, bool rotate_bands, hipStream_t stream)
{
    const int rotate = rotate_bands ? 1 : 0;
    if (n_frames <= 0 || max_w <= 0 || max_h <= 0) return;
    const bool fast = ((int64_t)H + 2) * W * 4 < ((int64_t)1 << 31) && hi_bounds_ok(0, W, 0, H) && max_w < (1 << 28);
    if (fast) {
        const int NW = nw == 1 ? 1 : (nw == 2 ? 2 : (nw >= 8 ? 8 : 4));        // windows per wave (measured on C2, 1 -> 2 -> 4: 0.198 -> 0.177 -> 0.171 ms; round 3, 4 -> 8: 0.162 -> 0.153, one source per frame 0.219 -> 0.199)
        const int nx = 1 << xcc_log2, chunks = (max_w + 256 * NW - 1) / (256 * NW), gpx = ((max_h + 3) / 4 + nx - 1) / nx;
        dim3 grid((unsigned)chunks * (unsigned)gpx * (unsigned)nx * (unsigned)n_frames);
#define HG_GEO(K) do { if (NW == 1) hipLaunchKernelGGL((k_geo_fast<K, 1>), grid, dim3(64, 4), 0, stream, frames, mats, img, W, H, n_imgs, img_stride, out, plain, xcc_log2, gpx, chunks, rotate); \
                       else if (NW == 4) hipLaunchKernelGGL((k_geo_fast<K, 4>), grid, dim3(64, 4), 0, stream, frames, mats, img, W, H, n_imgs, img_stride, out, plain, xcc_log2, gpx, chunks, rotate); \
                       else if (NW == 8) hipLaunchKernelGGL((k_geo_fast<K, 8>), grid, dim3(64, 4), 0, stream, frames, mats, img, W, H, n_imgs, img_stride, out, plain, xcc_log2, gpx, chunks, rotate); \
                       else hipLaunchKernelGGL((k_geo_fast<K, 2>), grid, dim3(64, 4), 0, stream, frames, mats, img, W, H, n_imgs, img_stride, out, plain, xcc_log2, gpx, chunks, rotate); } while (0)
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

} // namespace hg
