// experiments/hg_ablate.h -- timing ablations of the warp kernels (EXPERIMENTS.md).  NOT part of libhgwarp.so: this header is
// included only when hg_k_piecewise.hip is compiled with -DHG_EXPERIMENTS (`make experiments` -> lib/libhgwarp_exp.so, which
// no binding ever loads).  Every policy below makes the kernel write WRONG pixels by design; they exist to measure what a
// stage of the kernel costs.  Switches: HG_ABLATE (k_pw_rows) / HG_ABLATE_TRI (k_tri_spans) environment variables, bit sets:
//    1  span-list entries of 8 fixed L2-resident groups of the XCD band (with the row's own count): prologue at cache-hit cost
//    2  no gathers (the byte offset is stored instead of the pixel)
//    4  no stores
//    8  no triangle search (every pixel takes some span of the row)
//   16  gather addresses of a 16 px x 4 row lane patch instead of 64 px x 1 row
//   32  k_tri_spans: no slot atomics            64  k_tri_spans: no entry stores
#pragma once

namespace hg {

template <int ABL>
struct Ablate : NoExperiment {
    __device__ static __forceinline__ size_t list_base(size_t own, int xcd, int groups_per_xcd, int bi, int f, int rows_per_group, int row, int cap)
    {
        return (ABL & 1) ? ((size_t)(xcd * groups_per_xcd + ((bi - f * groups_per_xcd) & 7)) * rows_per_group + row) * cap : own;
    }
    __device__ static __forceinline__ bool fake_search(int *best, unsigned long long &any, int base, int w, int cnt)
    {
        if (!(ABL & 8)) return false;
        any = 1ull;
        best[0] = best[1] = best[2] = best[3] = (base + w % (cnt > 0 ? cnt : 1)) * 48;
        return true;
    }
    __device__ static __forceinline__ double pixel_x(double xd, int c0, int lane, int k, int x_off)
    {
        return (ABL & 16) ? (double)(c0 + (lane & 15) + 16 * k + x_off) : xd;
    }
    __device__ static __forceinline__ double pixel_hy(double hy, int lane) { return (ABL & 16) ? hy + (double)(lane >> 4) : hy; }
    __device__ static __forceinline__ uint32_t gather(__amdgpu_buffer_rsrc_t src, uint32_t off)
    {
        return (ABL & 2) ? off : __builtin_amdgcn_raw_buffer_load_b32(src, off, 0, 0);
    }
    __device__ static __forceinline__ bool skip_store(const uint32_t *px) { return (ABL & 4) && (px[0] ^ px[1] ^ px[2] ^ px[3]) != 0x9e3779b9u; }
    __device__ static __forceinline__ int slot(int32_t *cnt, int t, int64_t y) { return (ABL & 32) ? (int)((t * 7 + (int)y) & 31) : atomicAdd(cnt, 1); }
    static constexpr bool store_entries = !(ABL & 64);
    static constexpr bool tri_solve = !(ABL & 128);          // k_tri_spans_grouped: skip the per-triangle solves
};

static bool launch_tri_spans_ablated(const PwMesh &mesh, const PwFrames &fr, const RowLists &rl, dim3 grid, dim3 block, hipStream_t stream)
{
    static const int abl = getenv("HG_ABLATE_TRI") ? atoi(getenv("HG_ABLATE_TRI")) : 0;
    switch (abl) {
    case 32: hipLaunchKernelGGL((k_tri_spans<Ablate<32>, false>), grid, block, 0, stream, mesh, fr, rl); return true;
    case 64: hipLaunchKernelGGL((k_tri_spans<Ablate<64>, false>), grid, block, 0, stream, mesh, fr, rl); return true;
    case 96: hipLaunchKernelGGL((k_tri_spans<Ablate<96>, false>), grid, block, 0, stream, mesh, fr, rl); return true;
    default: return false;
    }
}

static bool launch_pw_rows_ablated(const PwMesh &mesh, const PwFrames &fr, const RowLists &rl, uint8_t *out, int16_t *map_out, int rpx, int rg,
                                   int32_t *status_next, dim3 grid, hipStream_t stream)
{
    static const int abl = getenv("HG_ABLATE") ? atoi(getenv("HG_ABLATE")) : 0;
#define HG_ABL(N) case N: hipLaunchKernelGGL((k_pw_rows<kRowSpanCapFast, Ablate<N>, false>), grid, dim3(256), 0, stream, mesh, fr, rl, out, map_out, rpx, rg, status_next); return true
    switch (abl) {
    HG_ABL(1); HG_ABL(2); HG_ABL(4); HG_ABL(6); HG_ABL(8); HG_ABL(14); HG_ABL(16);
    default: return false;
    }
#undef HG_ABL
}

} // namespace hg
