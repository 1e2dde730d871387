// hg_dev.h -- device-side helpers shared by the kernel translation units (hg_k_*.hip): rounding blocks, span updates,
// the slow-path pixel body, buffer policies, and the experiment hooks' product policy.  gfx950 only.
// Citations are file:line into the reference's Homography.js (v1.8.0).  Design notes: DESIGN.md §4.
#pragma once
#include "hg_kernels.h"
#include <cstdlib>
#include <type_traits>

namespace hg {

// A kernel flags frame f (status word in device memory, read by hg_sync only when the host-visible flag says there is something to read)
__device__ __forceinline__ void flag_frame(const PwFrames &fr, int f, int32_t bits)
{
    atomicOr(&fr.status[f], bits);
    if (fr.host_flag) __hip_atomic_store(fr.host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}



// A dword gather by ELEMENT index: `buffer_load_dword ... idxen` through a descriptor of stride 4 (address = base + index * 4; out of range
// -- 0, the JS `undefined -> 0` -- when index >= NUM_RECORDS, counted in elements).  Saves the `<< 2` of the byte-offset form in every pixel
// of the kernels that run on their vector port.  clang has no builtin for the structured form; this is the LLVM intrinsic itself.
extern "C" __device__ uint32_t hg_struct_load_u32(__amdgpu_buffer_rsrc_t rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.ptr.buffer.load.i32");

// Which frame and which row group of it the `bi`-th workgroup of XCD `xcd` takes (k_pw_rows / k_pw_patch / k_pw_tile); false: padding.
//   * plain: the XCD walks its contiguous band of `groups_per_xcd` groups frame by frame (band xcd, or xcd + frame where the bands rotate);
//   * fr.sub_groups > 0 (shared source, fixed bands; PwFrames): the frame's groups are cut into sub-bands of sub_groups groups, sub-band j belongs
//     to XCD j mod XCCs, and the XCD takes all frames of one of its sub-bands before the next.
__device__ __forceinline__ bool frame_group(const PwFrames &fr, int xcd, int bi, int groups_per_xcd, int &f, int &g)
{
    if (fr.sub_groups > 0) {
        const int per_sub = fr.sub_groups * fr.n_frames, sb = bi / per_sub, rem = bi - sb * per_sub;
        f = rem / fr.sub_groups;
        g = ((sb << fr.xcc_log2) + xcd) * fr.sub_groups + (rem - f * fr.sub_groups);
        return g < (groups_per_xcd << fr.xcc_log2);
    }
    f = bi / groups_per_xcd;
    g = ((xcd + (fr.xcc_rotate ? f : 0)) & ((1 << fr.xcc_log2) - 1)) * groups_per_xcd + (bi - f * groups_per_xcd);
    return true;
}
// (launchers: groups per sub-band and the padded number of group slots per XCD and frame)
inline int sub_groups_of(const PwFrames &fr, int groups_per_xcd) { return fr.sub_bands > 1 ? (groups_per_xcd + fr.sub_bands - 1) / fr.sub_bands : 0; }
inline int padded_groups(int groups_per_xcd, int sub_groups) { return sub_groups > 0 ? ((groups_per_xcd + sub_groups - 1) / sub_groups) * sub_groups : groups_per_xcd; }

// ------------------------------------------------------------------------------------------------ bounds on the high dwords
// The bounds tests :1047 / :1001 are made on h = RTN(s + 0.5):  a <= s < b  <=>  a + 0.5 <= h < b + 0.5  (a, b integers).
// When 0 <= a and b < 2^20, both limits are doubles >= 0.5 whose LOW dword is zero (at most 21 significant bits), and then
// for EVERY bit pattern of h
//        lo <= h < hi   <=>   (uint32)(hi32(h) - hi32(lo)) < hi32(hi) - hi32(lo)
// -- non-negative doubles order like their bit patterns, and with a zero low dword of the limit the 64-bit comparison is
// decided by the high dwords alone; patterns with the sign bit set (negative, -0, negative NaNs) have hi32(h) >= 2^31 >
// hi32(hi) and positive NaNs / +Inf have hi32(h) >= 0x7ff00000 > hi32(hi): both fail, as the fp64 compares do.  One 32-bit
// subtract and one 32-bit compare per coordinate instead of two fp64 compares (fp64 instructions issue at half the rate).
struct HiBounds { uint32_t lox, rx, loy, ry; };
__host__ __device__ __forceinline__ bool hi_bounds_ok(int64_t ax, int64_t bx, int64_t ay, int64_t by)      // limits a + 0.5, b + 0.5
{
    return ax >= 0 && ay >= 0 && bx < (1 << 20) && by < (1 << 20) && ax <= bx && ay <= by;
}
__device__ __forceinline__ HiBounds make_hi_bounds(double lox, double hix, double loy, double hiy)       // wave-uniform: kept in SGPRs
{
    HiBounds b;
    b.lox = (uint32_t)__builtin_amdgcn_readfirstlane(__double2hiint(lox)); b.rx = (uint32_t)__builtin_amdgcn_readfirstlane(__double2hiint(hix)) - b.lox;
    b.loy = (uint32_t)__builtin_amdgcn_readfirstlane(__double2hiint(loy)); b.ry = (uint32_t)__builtin_amdgcn_readfirstlane(__double2hiint(hiy)) - b.loy;
    return b;
}
__device__ __forceinline__ bool hi_inb(const HiBounds &b, double hx, double hy)
{
    return (int)(((uint32_t)__double2hiint(hx) - b.lox) < b.rx) & (int)(((uint32_t)__double2hiint(hy) - b.loy) < b.ry);
}

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

// For 8 doubles, in place: h[i] = RTN(h[i] + 0.5), then r[i] = RTN(h[i] + M), M = 1.5 * 2^52, with the fp64 rounding mode
// switched to round-toward-minus-infinity for exactly these 16 adds.  floor(h) == floor(v + 0.5 exactly) == Math.round(v)
// for every finite double (RTN never crosses an integer upward; this also gets 0.49999999999999994 right), and it appears
// as the low dword of r.  h itself serves the bounds test:  a <= v < b  <=>  a + 0.5 <= h < b + 0.5  (a, b integers).
// (h is an in/out operand so that the 8 inputs and the 8 h share registers: 32 VGPRs for the block instead of 48.)
// (NOT `asm volatile`: the block is a pure function of its operands -- it switches the rounding mode and switches it back -- and without the
//  qualifier the compiler may move the NEXT window's arithmetic across it, ahead of this window's gathers: k_geo_fast with one source per
//  frame - 4.6 % on two boxes, nothing with a shared source, EXPERIMENTS.md R6.16; the other blocks below measured the same either way)
__device__ __forceinline__ void round_x8(double h[8], double r[8])
{
    const double M = 6755399441055744.0;
    asm(
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

// Math.round of four values KNOWN to lie inside the source window (|v| < 2^30) with ONE add each: r = RTN(v + (1.5 * 2^51 + 0.5)).
// In [2^51, 2^52) a double's ulp is 0.5, so the constant is exact and r = 1.5 * 2^51 + (v + 0.5 rounded DOWN to the half grid): its low
// dword holds floor(2 v + 1), and Math.round(v) = floor(v + 0.5) = that >> 1 (arithmetic: negative values floor too).  Used by the
// pixel loops for windows whose spans are all flagged "wholly in bounds" (no bounds test, so no separate h = RTN(v + 0.5) is needed).
__device__ __forceinline__ void round_half_x4(const double v[4], int r[4])
{
    const double C = 3377699720527872.5;
    double t0, t1, t2, t3;
    asm volatile(
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
        "v_add_f64 %0, %4, %8\n\t" "v_add_f64 %1, %5, %8\n\t" "v_add_f64 %2, %6, %8\n\t" "v_add_f64 %3, %7, %8\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(C));
    r[0] = __double2loint(t0) >> 1; r[1] = __double2loint(t1) >> 1; r[2] = __double2loint(t2) >> 1; r[3] = __double2loint(t3) >> 1;
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

// same with the span's length and key in SCALAR registers (broadcast out of a lane by v_readlane: no LDS round trip per span)
__device__ __forceinline__ void span_max4s(int best[4], int d, int len, int key)
{
    const int d1 = d + 64, d2 = d + 128, d3 = d + 192;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_gt_u32_e32 vcc, %[len], %[d0]\n\t" "v_max_i32_e32 %[b0], %[key], %[b0]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_gt_u32_e32 vcc, %[len], %[d1]\n\t" "v_max_i32_e32 %[b1], %[key], %[b1]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_gt_u32_e32 vcc, %[len], %[d2]\n\t" "v_max_i32_e32 %[b2], %[key], %[b2]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_gt_u32_e32 vcc, %[len], %[d3]\n\t" "v_max_i32_e32 %[b3], %[key], %[b3]\n\t" "s_mov_b64 exec, %[sv]"
        : [b0] "+v"(best[0]), [b1] "+v"(best[1]), [b2] "+v"(best[2]), [b3] "+v"(best[3]), [sv] "=&s"(sv)
        : [d0] "v"(d), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [len] "s"(len), [key] "s"(key)
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

} // namespace hg
