// calib_fetch.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS library's access patterns.
//
// MI355X_MICROARCH.md (HBM section) calibrates FETCH_SIZE only for 16 B/lane streaming loads (x2) and says every other
// width is uncalibrated.  The warp kernels gather with 4 B/lane `buffer_load_dword` and store with 4 B/lane non-temporal
// `buffer_store_dword`, so this tool runs kernels of KNOWN byte counts in exactly those instruction forms over a buffer
// far larger than L2 + the 256 MiB Infinity Cache (default 2 GiB, every byte touched once per kernel):
//     calib_load16        16 B/lane global_load_dwordx4, fully coalesced            -> N bytes read
//     calib_load4         4 B/lane buffer_load_dword, 64 consecutive dwords / wave  -> N bytes read   (k_pw_rows' gather, no shear)
//     calib_load4_line    4 B/lane, every lane on its own 128-B line (1 dword used) -> N/32 useful, N/32 * 32 = N fetched if lines are 128 B
//     calib_load4_half    4 B/lane, every lane on its own 64-B half line            -> tells 64-B from 128-B request granularity
//     calib_store16       16 B/lane global_store_dwordx4                            -> N bytes written
//     calib_store4_nt     4 B/lane non-temporal buffer_store_dword, 256 B / wave    -> N bytes written (k_pw_rows' store)
// Run under `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE` (tools/calibrate_pmc.sh); the ratio
// known bytes / counter is the correction factor recorded in profiles/hbm_traffic.json.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void calib_load16(const uint4 *__restrict__ p, size_t n16, uint32_t *sink)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (; i < n16; i += stride) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x9e3779b9u) *sink = acc;
}

// every wave instruction reads 64 consecutive dwords (256 B); the buffer is walked in 2 GiB-safe chunks of 1 GiB descriptors
__global__ __launch_bounds__(256) void calib_load4(const uint8_t *__restrict__ p, size_t bytes, uint32_t *sink)
{
    uint32_t acc = 0;
    const size_t chunk = (size_t)1 << 30;
    for (size_t base = 0; base < bytes; base += chunk) {
        const uint32_t len = (uint32_t)((bytes - base) < chunk ? (bytes - base) : chunk);
        const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p + base), 0, len, 0x00020000);
        for (uint32_t off = (blockIdx.x * 256u + threadIdx.x) * 4u; off < len; off += gridDim.x * 1024u)
            acc ^= __builtin_amdgcn_raw_buffer_load_b32(src, off, 0, 0);
    }
    if (acc == 0x9e3779b9u) *sink = acc;
}

// every lane reads ONE dword of its own `unit`-byte block (unit = 128: one per cache line; 64: one per half line)
template <int UNIT>
__global__ __launch_bounds__(256) void calib_load4_sparse(const uint8_t *__restrict__ p, size_t bytes, uint32_t *sink)
{
    uint32_t acc = 0;
    const size_t chunk = (size_t)1 << 30;
    for (size_t base = 0; base < bytes; base += chunk) {
        const uint32_t len = (uint32_t)((bytes - base) < chunk ? (bytes - base) : chunk);
        const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p + base), 0, len, 0x00020000);
        const uint32_t units = len / UNIT;
        for (uint32_t u = blockIdx.x * 256u + threadIdx.x; u < units; u += gridDim.x * 256u)
            acc ^= __builtin_amdgcn_raw_buffer_load_b32(src, u * (uint32_t)UNIT + ((u * 4u) & (UNIT - 1)), 0, 0);
    }
    if (acc == 0x9e3779b9u) *sink = acc;
}

__global__ __launch_bounds__(256) void calib_store16(uint4 *__restrict__ p, size_t n16)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n16; i += stride) p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

__global__ __launch_bounds__(256) void calib_store4_nt(uint8_t *__restrict__ p, size_t bytes)
{
    const size_t chunk = (size_t)1 << 30;
    for (size_t base = 0; base < bytes; base += chunk) {
        const uint32_t len = (uint32_t)((bytes - base) < chunk ? (bytes - base) : chunk);
        const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(p + base, 0, len, 0x00020000);
        for (uint32_t off = (blockIdx.x * 256u + threadIdx.x) * 4u; off < len; off += gridDim.x * 1024u)
            __builtin_amdgcn_raw_buffer_store_b32(off, dst, off, 0, 2 /* nt */);
    }
}

// mixed read + write streams (what a warp kernel with one source per frame does): N/2 bytes read, N/2 bytes written
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void calib_copy16(const v4u *__restrict__ src, v4u *__restrict__ dst, size_t n16)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n16; i += stride) {
        const v4u v = src[i];
        if (NT) __builtin_nontemporal_store(v, &dst[i]); else dst[i] = v;
    }
}

// 4 B/lane buffer loads + 4 B/lane non-temporal buffer stores, U independent loads in flight per lane
template <int U>
__global__ __launch_bounds__(256) void calib_copy4_nt(const uint8_t *__restrict__ s, uint8_t *__restrict__ d, uint32_t len)
{
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(s), 0, len, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(d, 0, len, 0x00020000);
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u;
    for (uint32_t base = wave * (U * 256u); base < len; base += nwaves * (U * 256u)) {
        uint32_t v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b32(src, base + k * 256u + lane * 4u, 0, 0);
#pragma unroll
        for (int k = 0; k < U; k++) __builtin_amdgcn_raw_buffer_store_b32(v[k], dst, base + k * 256u + lane * 4u, 0, 2);
    }
}

int main(int argc, char **argv)
{
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : (size_t)2048) << 20;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    uint8_t *buf = nullptr; uint32_t *sink = nullptr;
    CK(hipMalloc((void **)&buf, bytes));
    CK(hipMalloc((void **)&sink, 4));
    CK(hipMemset(buf, 0x5a, bytes));
    CK(hipDeviceSynchronize());
    const dim3 grid(256 * 16), block(256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char *name; int id; } tests[] = { {"calib_load16", 0}, {"calib_load4", 1}, {"calib_load4_line", 2}, {"calib_load4_half", 3},
                                                     {"calib_store16", 4}, {"calib_store4_nt", 5},
                                                     {"calib_copy16", 6}, {"calib_copy16_nt", 7}, {"calib_copy4_nt_x4", 8}, {"calib_copy4_nt_x16", 9} };
    for (auto &t : tests) {
        float best = 1e30f;
        for (int r = 0; r < reps; r++) {
            CK(hipEventRecord(e0));
            switch (t.id) {
            case 0: hipLaunchKernelGGL(calib_load16, grid, block, 0, 0, (const uint4 *)buf, bytes / 16, sink); break;
            case 1: hipLaunchKernelGGL(calib_load4, grid, block, 0, 0, buf, bytes, sink); break;
            case 2: hipLaunchKernelGGL(calib_load4_sparse<128>, grid, block, 0, 0, buf, bytes, sink); break;
            case 3: hipLaunchKernelGGL(calib_load4_sparse<64>, grid, block, 0, 0, buf, bytes, sink); break;
            case 4: hipLaunchKernelGGL(calib_store16, grid, block, 0, 0, (uint4 *)buf, bytes / 16); break;
            case 5: hipLaunchKernelGGL(calib_store4_nt, grid, block, 0, 0, buf, bytes); break;
            case 6: hipLaunchKernelGGL(calib_copy16<false>, grid, block, 0, 0, (const v4u *)buf, (v4u *)(buf + bytes / 2), bytes / 32); break;
            case 7: hipLaunchKernelGGL(calib_copy16<true>, grid, block, 0, 0, (const v4u *)buf, (v4u *)(buf + bytes / 2), bytes / 32); break;
            case 8: hipLaunchKernelGGL(calib_copy4_nt<4>, grid, block, 0, 0, buf, buf + bytes / 2, (uint32_t)(bytes / 2)); break;
            case 9: hipLaunchKernelGGL(calib_copy4_nt<16>, grid, block, 0, 0, buf, buf + bytes / 2, (uint32_t)(bytes / 2)); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double useful = t.id == 2 ? bytes / 32.0 : (t.id == 3 ? bytes / 16.0 : (double)bytes);
        printf("{\"kernel\": \"%s\", \"buffer_bytes\": %zu, \"useful_bytes\": %.0f, \"best_ms\": %.4f, \"useful_GBps\": %.1f, \"buffer_GBps\": %.1f}\n",
               t.name, bytes, useful, best, useful / best / 1e6, bytes / (double)best / 1e6);
    }
    CK(hipFree(buf)); CK(hipFree(sink));
    return 0;
}
