// hg_k_map.hip -- materialised triangle-map path: k_map_fill (atomicMax rasteriser), k_pw_from_map; the parity tap and the exact fallback
// Hand-written HIP for gfx950 (MI355X / CDNA4), wave64; fp64 coordinate math with contraction off so that nearest-neighbour
// source selection is bit-identical to the reference's JS doubles.
// Citations are file:line into the reference's Homography.js (v1.8.0).  Design notes: DESIGN.md §4.
#include "hg_dev.h"

namespace hg {

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

// Largest Int16Array VALUE among the first n cells (-1 when none is set): the reference-state warps compare it with the number of
// matrices they were given before any pixel indexes one (hg_warp_*_piecewise_state).
__global__ void k_map_max_i16(const int32_t *__restrict__ m32, size_t n, int32_t *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int best = -1;
    for (; i < n; i += stride) { const int v = (int)(int16_t)m32[i]; best = v > best ? v : best; }
    for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(best, o); best = w > best ? w : best; }
    if ((threadIdx.x & 63) == 0 && best >= 0) atomicMax(out, best);
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

void launch_fill_i32(int32_t *p, size_t n, int32_t v, hipStream_t stream)
{
    if (n == 0) return;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_fill_i32, dim3(blocks), dim3(256), 0, stream, p, n, v);
}

void launch_map_build(const PwMesh &mesh, const PwFrames &fr, int f, const FrameDesc &fd, int32_t *map32, hipStream_t stream)
{
    const size_t n = (fd.obj_w > 0 && fd.obj_h > 0) ? (size_t)fd.obj_w * fd.obj_h : 0;
    if (n == 0) return;
    launch_fill_i32(map32, n, -1, stream);                   // :850
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

void launch_map_max_i16(const int32_t *map32, size_t n, int32_t *out, hipStream_t stream)
{
    (void)hipMemsetAsync(out, 0xff, sizeof(int32_t), stream);      // -1
    if (n == 0) return;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_map_max_i16, dim3(blocks), dim3(256), 0, stream, map32, n, out);
}

void launch_map_to_i16(const int32_t *map32, int16_t *map16, size_t n, hipStream_t stream)
{
    if (n == 0) return;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_map_to_i16, dim3(blocks), dim3(256), 0, stream, map32, map16, n);
}

} // namespace hg
