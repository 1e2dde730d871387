// calib_shape.hip -- what an IMAGE-SHAPED mixed read/write stream reaches on MI355X, by traversal shape (round 3).
//
// The warp kernels with one source per frame move 4 B read + 4 B written per output pixel; a linear streaming copy in the
// same instruction forms reaches 4.5 (4 + 4 requests in flight per wave) .. 5.8 TB/s (16 + 16), tools/calib_fetch.  This
// tool replaces the linear walk by the walks a warp kernel can actually do over F frames of W x H RGBA8, each with its own
// source, under a sheared inverse map  src(x, y) = (x, y + floor(slope * x) + dy0)  (slope 0.026 ~ C3, 0.5 ~ C4 / C5):
//   rows   k_pw_rows' walk: workgroup = 4 consecutive output rows, wave j walks row r0 + j window by window (256 px), lane l
//          owns pixels c0 + l + 64k; PH windows' gathers (4 B/lane buffer loads) are issued before their 4*PH nt stores.
//   tile   LDS-staged walk: workgroup = TH output rows x full width, column tiles of TW pixels; the tile's source bounding
//          box is loaded row by row with 16 B/lane loads into LDS (double buffered), pixels are picked from LDS, every lane
//          stores 4 consecutive pixels with ONE 16 B nt store.
// ALU padding: `pad` dependent fp64 fma per pixel stand in for the transform / rounding / bounds arithmetic.
// Prints GB/s of (4 B read + 4 B written) per output pixel.  Not on the product path.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <utility>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

struct Shape { int W, H, F; float slope; int dy0; int pad; };

__device__ __forceinline__ double pad_fma(double v, int pad)
{
    for (int i = 0; i < pad; i++) v = fma(v, 1.0000001, 0.25);
    return v;
}

// ---------------------------------------------------------------------------------------------- rows (k_pw_rows' walk)
template <int PH>
__global__ __launch_bounds__(256) void k_rows(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, Shape s, int groups_per_xcd)
{
    const int bid = blockIdx.x, xcd = bid & 7, bi = bid >> 3;
    const int f = bi / groups_per_xcd;
    const int r0 = (xcd * groups_per_xcd + (bi - f * groups_per_xcd)) * 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = r0 + wave;
    if (r >= s.H) return;
    const size_t fbytes = (size_t)s.W * s.H * 4;
    const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src + (size_t)f * fbytes), 0, (int)fbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t db = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)f * fbytes + (size_t)r * s.W * 4, 0, s.W * 4, 0x00020000);
    const int nwin = (s.W + 255) >> 8;
    for (int wb = 0; wb < nwin; wb += PH) {
        uint32_t px[PH][4];
#pragma unroll
        for (int p = 0; p < PH; p++) {
            const int c0 = (wb + p) << 8;
            if (wb + p >= nwin) break;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = c0 + lane + 64 * k;
                double yy = (double)r + floor((double)s.slope * (double)x) + (double)s.dy0;
                yy = pad_fma(yy, s.pad) ;
                int sy = (int)yy;
                sy = sy < 0 ? 0 : (sy >= s.H ? s.H - 1 : sy);
                const uint32_t off = x < s.W ? (uint32_t)(sy * s.W + x) * 4u : 0xffffffffu;
                px[p][k] = __builtin_amdgcn_raw_buffer_load_b32(sb, off, 0, 0);
            }
        }
#pragma unroll
        for (int p = 0; p < PH; p++) {
            const int c0 = (wb + p) << 8;
            if (wb + p >= nwin) break;
#pragma unroll
            for (int k = 0; k < 4; k++) __builtin_amdgcn_raw_buffer_store_b32(px[p][k], db, (c0 + lane + 64 * k) * 4, 0, 2);
        }
    }
}

// ---------------------------------------------------------------------------------------------- tile (LDS-staged walk)
// Workgroup: TH output rows, all column tiles of TW pixels.  Footprint of a tile: source rows [r0 + ymin, r0 + TH + ymax],
// columns [x0, x0 + TW): FR x TW pixels, FR = TH + ceil(|slope| * TW) + 1.  Thread t owns pixels (row t / (TW/4), 4 px at
// (t % (TW/4)) * 4) [+ more rows when TH * TW / 4 > 256].
template <int TW, int TH, int FRMAX>
__global__ __launch_bounds__(256) void k_tile(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, Shape s, int groups_per_xcd)
{
    extern __shared__ __align__(16) uint32_t lds[];           // 2 buffers of FRMAX x TW dwords
    const int bid = blockIdx.x, xcd = bid & 7, bi = bid >> 3;
    const int f = bi / groups_per_xcd;
    const int r0 = (xcd * groups_per_xcd + (bi - f * groups_per_xcd)) * TH;
    if (r0 >= s.H) return;
    const size_t fbytes = (size_t)s.W * s.H * 4;
    const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src + (size_t)f * fbytes), 0, (int)fbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t db = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)f * fbytes, 0, (int)fbytes, 0x00020000);
    const int ntile = (s.W + TW - 1) / TW;
    constexpr int QPR = TW / 4;                                // 16-byte quads per tile row
    const int tid = threadIdx.x;
    auto foot = [&](int t, int &y0, int &fr) {                 // source rows of tile t: [y0, y0 + fr)
        const int x0 = t * TW, x1 = x0 + TW - 1;
        const int a = (int)floorf(s.slope * (float)x0), b = (int)floorf(s.slope * (float)x1);
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        y0 = r0 + lo + s.dy0; fr = TH + (hi - lo);
        if (fr > FRMAX) fr = FRMAX;
    };
    constexpr int NREG = (FRMAX * QPR + 255) / 256;
    auto issue = [&](int t, v4u *regs) {                       // 16 B/lane loads of the footprint -> registers (then ds_write)
        int y0, fr; foot(t, y0, fr);
#pragma unroll
        for (int i = 0; i < NREG; i++) {
            const int q = tid + i * 256;
            if (q < fr * QPR) {
                const int row = q / QPR, qc = q - row * QPR;
                int sy = y0 + row; sy = sy < 0 ? 0 : (sy >= s.H ? s.H - 1 : sy);
                const int x = t * TW + qc * 4;
                const uint32_t off = x < s.W ? (uint32_t)(sy * s.W + x) * 4u : 0xffffffffu;
                regs[i] = __builtin_amdgcn_raw_buffer_load_b128(sb, off, 0, 0);
            }
        }
    };
    auto land = [&](int t, int buf, const v4u *regs) {
        int y0, fr; foot(t, y0, fr);
#pragma unroll
        for (int i = 0; i < NREG; i++) {
            const int q = tid + i * 256;
            if (q < fr * QPR) *reinterpret_cast<v4u *>(lds + buf * (FRMAX * TW) + q * 4) = regs[i];
        }
    };
    v4u regs[NREG];
    issue(0, regs);
    land(0, 0, regs);
    __syncthreads();
    for (int t = 0; t < ntile; t++) {
        const int buf = t & 1;
        if (t + 1 < ntile) issue(t + 1, regs);  // next footprint in flight while this tile is resolved
        int y0, fr; foot(t, y0, fr);
        const uint32_t *fp = lds + buf * (FRMAX * TW);
        for (int q = tid; q < TH * QPR; q += 256) {
            const int row = q / QPR, qc = q - row * QPR;
            v4u o;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = t * TW + qc * 4 + k;
                double yy = (double)(r0 + row) + floor((double)s.slope * (double)x) + (double)s.dy0;
                yy = pad_fma(yy, s.pad);
                int fy = (int)yy - y0;
                fy = fy < 0 ? 0 : (fy >= fr ? fr - 1 : fy);
                o[k] = fp[fy * TW + qc * 4 + k];
            }
            const int x = t * TW + qc * 4;
            const uint32_t off = (x < s.W && r0 + row < s.H) ? (uint32_t)((r0 + row) * s.W + x) * 4u : 0xffffffffu;
            __builtin_amdgcn_raw_buffer_store_b128(o, db, off, 0, 2);
        }
        if (t + 1 < ntile) land(t + 1, buf ^ 1, regs);
        __syncthreads();
    }
}

int main(int argc, char **argv)
{
    Shape s; s.W = 3840; s.H = 2160; s.F = 64; s.slope = 0.026f; s.dy0 = -40; s.pad = 8;
    int reps = 5, ldsk = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--slope")) s.slope = (float)atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--pad")) s.pad = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--frames")) s.F = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--lds")) ldsk = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--w")) s.W = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--h")) s.H = atoi(argv[i + 1]);
    }
    const size_t fbytes = (size_t)s.W * s.H * 4, total = fbytes * s.F;
    uint8_t *src = nullptr, *dst = nullptr;
    CK(hipMalloc((void **)&src, total)); CK(hipMalloc((void **)&dst, total));
    CK(hipMemset(src, 0x5a, total)); CK(hipMemset(dst, 0, total));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct T { const char *name; int id; } tests[] = {
        {"rows_ph1", 0}, {"rows_ph2", 1}, {"rows_ph4", 2},
        {"tile_64x16", 3}, {"tile_128x8", 4}, {"tile_256x4", 5}, {"tile_128x16", 6}, {"tile_256x8", 7}, {"tile_64x32", 8},
    };
    for (auto &t : tests) {
        if (ldsk > 0 && t.id > 2) continue;
        float best = 1e30f;
        bool skipped = false;
        for (int r = 0; r < reps + 1; r++) {
            CK(hipEventRecord(e0));
            const int absl = (int)(fabsf(s.slope) * 256) + 2;
            auto rows_grid = [&](int rg) { const int gpx = ((s.H + rg - 1) / rg + 7) / 8; return std::pair<int, int>(gpx * 8 * s.F, gpx); };
#define TILE(TW, TH, FR) do { if ((int)(fabsf(s.slope) * TW) + TH + 1 > FR) { skipped = true; break; } auto g = rows_grid(TH); \
            hipLaunchKernelGGL((k_tile<TW, TH, FR>), dim3(g.first), dim3(256), 2 * FR * TW * 4, 0, src, dst, s, g.second); } while (0)
            switch (t.id) {
            case 0: { auto g = rows_grid(4); hipLaunchKernelGGL(k_rows<1>, dim3(g.first), dim3(256), ldsk * 1024, 0, src, dst, s, g.second); } break;
            case 1: { auto g = rows_grid(4); hipLaunchKernelGGL(k_rows<2>, dim3(g.first), dim3(256), ldsk * 1024, 0, src, dst, s, g.second); } break;
            case 2: { auto g = rows_grid(4); hipLaunchKernelGGL(k_rows<4>, dim3(g.first), dim3(256), ldsk * 1024, 0, src, dst, s, g.second); } break;
            case 3: if (absl < 100) TILE(64, 16, 24); else TILE(64, 16, 50); break;
            case 4: if (absl < 100) TILE(128, 8, 16); else TILE(128, 8, 74); break;
            case 5: if (absl < 100) TILE(256, 4, 12); else TILE(256, 4, 134); break;
            case 6: if (absl < 100) TILE(128, 16, 24); else TILE(128, 16, 82); break;
            case 7: if (absl < 100) TILE(256, 8, 16); else skipped = true; break;
            case 8: if (absl < 100) TILE(64, 32, 40); else TILE(64, 32, 66); break;
            }
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms < best) best = ms;
        }
        if (skipped) { printf("{\"kernel\": \"%s\", \"skipped\": true}\n", t.name); continue; }
        printf("{\"kernel\": \"%s\", \"slope\": %.3f, \"pad\": %d, \"lds\": %d, \"frames\": %d, \"best_ms\": %.4f, \"GBps\": %.1f}\n",
               t.name, s.slope, s.pad, ldsk, s.F, best, 2.0 * total / best / 1e6);
    }
    return 0;
}
