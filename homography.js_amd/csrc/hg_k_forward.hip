// hg_k_forward.hip -- forward (scatter-semantics) warps: scatter + gather, and the tile-binned gather kernels
// Hand-written HIP for gfx950 (MI355X / CDNA4), wave64; fp64 coordinate math with contraction off so that nearest-neighbour
// source selection is bit-identical to the reference's JS doubles.
// Citations are file:line into the reference's Homography.js (v1.8.0).  Design notes: DESIGN.md §4.
#include "hg_dev.h"

namespace hg {

// (forward tile kernels: flag frame f for the scatter + gather redo, and tell hg_sync through the host-visible word that there is something to read)
__device__ __forceinline__ void fwd_flag(const FwdPwTiles &p, int f, int32_t bits)
{
    atomicOr(&p.status[f], bits);
    if (p.host_flag) __hip_atomic_store(p.host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
__global__ __launch_bounds__(256) void k_fwd_tiles(FwdBatch batch, const uint8_t *__restrict__ img, int n_imgs, uint64_t img_stride, int W, int H, uint8_t *__restrict__ out)
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
    // (one source per frame -- the loop `for (f) warp(frame_f)` -- : frame f reads image f mod n_imgs, like the inverse kernels)
    const uint32_t *__restrict__ img32 = reinterpret_cast<const uint32_t *>(n_imgs > 1 ? img + (uint64_t)(f % n_imgs) * img_stride : img);
    uint32_t *__restrict__ o32 = reinterpret_cast<uint32_t *>(out + fd.out_off);
    const int cx = tid & (kFwdTileW - 1);
    for (int cy = tid >> 6; cy < ty1 - ty0; cy += 4) {
        if (tx0 + cx >= tx1) continue;
        const int w = s_win[cy * kFwdTileW + cx];
        o32[(size_t)(ty0 + cy) * fd.obj_w + tx0 + cx] = w >= 0 ? img32[w] : 0u;
    }
}

// ------------------------------------------------------------------------------------------------ forward piecewise, tile-binned
// inclusive prefix sum over the NW * 64 threads of a workgroup (two barriers; s_wsum: NW ints of LDS); total = the sum over all
template <int NW>
__device__ __forceinline__ int block_scan_incl(int v, int *s_wsum, int lane, int wave, int &total)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d); if (lane >= d) v += o; }
    __syncthreads();
    if (lane == 63) s_wsum[wave] = v;
    __syncthreads();
    total = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) { const int ws = s_wsum[w]; if (w < wave) v += ws; total += ws; }
    return v;
}
__device__ __forceinline__ int block_scan_incl(int v, int *s_wsum, int lane, int wave)
{
    int total;
    return block_scan_incl<4>(v, s_wsum, lane, wave, total);
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
    if (bad) { if (lane == 0) fwd_flag(p, f, FWD_FALLBACK); return; }
    const int64_t ua = (int64_t)floor(umin) - 2, ub = (int64_t)ceil(umax) + 2, va = (int64_t)floor(vmin) - 2, vb = (int64_t)ceil(vmax) + 2;
    const int64_t kmin = floordiv64(ua, fd.obj_w), kmax = floordiv64(ub, fd.obj_w);
    if (kmin < -2 || kmax > 2) { if (lane == 0) fwd_flag(p, f, FWD_FALLBACK); return; }
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
            else fwd_flag(p, f, FWD_OVERFLOW);
        }
    }
}

// One workgroup per 64 x 64 output tile: same three steps as k_fwd_tiles -- conservative candidates, the reference's exact
// arithmetic per candidate (:957-964), atomicMax of the raster rank on 4096 winner words in LDS -- with the candidates coming
// from the tile's (triangle, k) entries: (0) one lane per entry: the source rows of the tile rectangle's pre-image under that
// triangle's matrix, cut to the triangle's cell bbox; (1) one lane per (entry, row): the x interval; (2) one lane per source
// pixel, kept only if the forward map assigns it to this triangle.  Both levels are laid out back to back with prefix sums, so
// twenty small triangles cost what one large one costs.
//
// Shape of the workgroup (round 3; the loop was latency-bound, not ALU-bound: 46 of 65 us per 4K frame sat in step (2) with four
// waves per SIMD, each walking a prefix array to find its row): 512 threads share the tile's 16 KB of winner words (8 waves per
// SIMD at 4 workgroups per CU), and step (2) finds its row in ONE LDS read -- the rows of a pass are cut into 16-candidate
// segments listed in s_seg, candidate c belongs to segment c >> 4 -- then reads one 16-byte row record and the matrix.
constexpr int kPwT = 512;                  // threads per tile workgroup == (entry, row) records per pass
constexpr int kPwSegW = 16, kPwSegLog2 = 4;
constexpr int kPwSegCap = 1024;            // segments per round (16 384 candidates); a pass with more takes several rounds
constexpr int kPwUnroll = 2;               // candidates in flight per lane (4: 70 VGPRs, 7 waves per SIMD; 2 with the cap below: 8)
static_assert(kFwdPwCapMax <= kPwT && kPwT == 512, "row records carry the row's slot in 9 bits");

__global__ __launch_bounds__(kPwT) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fwd_pw_tiles(FwdPwTiles p, const uint8_t *__restrict__ img, int n_imgs, uint64_t img_stride, int W, int H, uint8_t *__restrict__ out)
{
    __shared__ uint32_t s_win[kFwdTileW * kFwdTileH];                  // 0 = no writer, else ((map row << 16) | map column) + 1
    __shared__ float s_m[kFwdPwCapMax][6];                             // (the matrices ARE floats: _trianglesTransforms is a Float32Array)
    __shared__ int s_etk[kFwdPwCapMax], s_eylo[kFwdPwCapMax], s_epre[kFwdPwCapMax];
    __shared__ int s_ebase[kFwdPwCapMax];                              // index of the entry's row-extent record for source row 0 (< 2^27 rows in total)
    __shared__ int4 s_row[kPwT];                                       // {first x, length, y, entry | t << 8 | (k + 2) << 24}
    __shared__ uint32_t s_seg[kPwSegCap];                              // row slot | (segment of that row) << 9
    __shared__ int s_wsum[kPwT / 64];
    const int f = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FrameDesc fd = p.frames[f];
    const int tx0 = blockIdx.x * kFwdTileW, ty0 = blockIdx.y * kFwdTileH;
    if (tx0 >= fd.obj_w || ty0 >= fd.obj_h) return;
    const size_t tidx = ((size_t)f * p.tsy + blockIdx.y) * p.tsx + blockIdx.x;
    if (p.status[f] != 0) { if (tid == 0) p.tile_cnt[tidx] = 0; return; }   // flagged by k_fwd_pw_bins: the host redoes this frame (counters are left zero for the next batch)
    const int tx1 = min(tx0 + kFwdTileW, fd.obj_w), ty1 = min(ty0 + kFwdTileH, fd.obj_h);
    for (int i = tid; i < kFwdTileW * kFwdTileH; i += kPwT) s_win[i] = 0u;
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
        for (int j = 0; j < 6; j++) { const float v = fwd[(size_t)t * 6 + j]; m[j] = v; s_m[tid][j] = v; }
        const int cy0 = p.bbox[4 * t + 1], cy1 = p.bbox[4 * t + 3];
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
        s_etk[tid] = t | ((k + 2) << 16); s_eylo[tid] = ylo;
        s_ebase[tid] = (int)((long long)p.rowoff[t] - (long long)(cy0 + p.min_src_y));
    }
    {
        int total;
        const int incl = block_scan_incl<kPwT / 64>(nrows, s_wsum, lane, wave, total);
        if (tid < kFwdPwCapMax) s_epre[tid] = incl;
        if (tid == 0) p.tile_cnt[tidx] = 0;                            // (every thread read it before the scan's barriers) zero for the next batch
    }
    __syncthreads();
    const int R = s_epre[kFwdPwCapMax - 1];

#pragma unroll 1
    for (int q0 = 0; q0 < R; q0 += kPwT) {
        // (1) one lane per (entry, source row)
        int len = 0, xa = 0, pe = 0, py = 0, etk = 0;
        const int q = q0 + tid;
        if (q < R) {
            int lo = 0, hi = E - 1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_epre[mid] > q) hi = mid; else lo = mid + 1; }
            pe = lo;
            py = s_eylo[pe] + (q - (pe ? s_epre[pe - 1] : 0));
            etk = s_etk[pe];
            const int k = (etk >> 16) - 2;
            const double m0 = s_m[pe][0], m1 = s_m[pe][1], m2 = s_m[pe][2], m3 = s_m[pe][3], m4 = s_m[pe][4], m5 = s_m[pe][5];
            const double fx_lo = (double)(tx0 + k * fd.obj_w) + fd.x_off - 0.5 - eps, fx_hi = (double)(tx1 - 1 + k * fd.obj_w) + fd.x_off + 0.5 + eps;
            const double fy_lo = (double)(ty0 - k) + fd.y_off - 0.5 - eps, fy_hi = (double)(ty1 - 1 - k) + fd.y_off + 0.5 + eps;
            const double yd = (double)py, cx = m2 * yd + m4, cy = m3 * yd + m5;
            const double a[4] = { m0, -m0, m1, -m1 }, b[4] = { cx - fx_lo, fx_hi - cx, cy - fy_lo, fy_hi - cy };
            const int2 ext = *reinterpret_cast<const int2 *>(p.rowext + 2 * ((long long)s_ebase[pe] + py));     // this triangle's cells in map row py
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
        const int nseg = (len + kPwSegW - 1) >> kPwSegLog2;
        int S;
        const int sp = block_scan_incl<kPwT / 64>(nseg, s_wsum, lane, wave, S) - nseg;    // (its first barrier also fences the previous pass's readers)
        s_row[tid] = make_int4(xa, len, py, pe | ((etk & 0xffff) << 8) | ((etk >> 16) << 24));
#pragma unroll 1
        for (int B = 0; B < S; B += kPwSegCap) {
            for (int j = max(0, B - sp); j < nseg && sp + j - B < kPwSegCap; j++) s_seg[sp + j - B] = (uint32_t)tid | ((uint32_t)j << 9);
            __syncthreads();
            // (2) one lane per candidate source pixel, kPwUnroll of them in flight
            const int C = min(S - B, kPwSegCap) << kPwSegLog2;
#pragma unroll 1
            for (int c0 = tid; c0 < C; c0 += kPwT * kPwUnroll) {
                int xs[kPwUnroll], ys[kPwUnroll], ws[kPwUnroll], owner[kPwUnroll];
#pragma unroll
                for (int j = 0; j < kPwUnroll; j++) {
                    const int c = c0 + kPwT * j;
                    ws[j] = -1; xs[j] = 0; ys[j] = 0;
                    if (c < C) {
                        const uint32_t sg = s_seg[c >> kPwSegLog2];
                        const int4 rr = s_row[sg & (kPwT - 1)];
                        const int off = (int)(sg >> 9) * kPwSegW + (c & (kPwSegW - 1));
                        if (off < rr.y) { xs[j] = rr.x + off; ys[j] = rr.z; ws[j] = rr.w; }
                    }
                }
#pragma unroll
                for (int j = 0; j < kPwUnroll; j++)
                    owner[j] = ws[j] >= 0 ? (int)(int16_t)p.fmap[(ys[j] - p.min_src_y) * p.map_w + (xs[j] - p.min_src_x)] : -1;     // (cell < 2^31: forward_limits)
#pragma unroll
                for (int j = 0; j < kPwUnroll; j++) {
                    if (ws[j] < 0) continue;
                    const int e = ws[j] & 0xff, t = (ws[j] >> 8) & 0xffff, k = (ws[j] >> 24) - 2;
                    const double m0 = s_m[e][0], m1 = s_m[e][1], m2 = s_m[e][2], m3 = s_m[e][3], m4 = s_m[e][4], m5 = s_m[e][5];
                    if (owner[j] != t) continue;                                                    // :957: this pixel uses another triangle's matrix
                    const double m[6] = { m0, m1, m2, m3, m4, m5 };
                    double nx, ny;
                    apply_affine(m, (double)xs[j], (double)ys[j], nx, ny);                          // :961
                    double uh = nx - (double)fd.x_off, vh = ny - (double)fd.y_off;                  // :962
                    int ui, vi;
                    round_x2(uh, vh, ui, vi);
                    if (!(fabs(uh) < 1.0e9 && fabs(vh) < 1.0e9)) continue;
                    const int col = ui - k * fd.obj_w, row = vi + k;
                    // winner = the LAST source pixel in raster order == the largest (row, column) key; map_w <= 65535 (launcher)
                    const uint32_t key = ((uint32_t)(ys[j] - p.min_src_y) << 16 | (uint32_t)(xs[j] - p.min_src_x)) + 1u;
                    if (row >= ty0 && row < ty1 && col >= tx0 && col < tx1) atomicMax(&s_win[(row - ty0) * kFwdTileW + (col - tx0)], key);
                }
            }
            __syncthreads();                                                                        // (s_seg / s_row are rewritten next)
        }
    }
    __syncthreads();
    const uint32_t *__restrict__ img32 = reinterpret_cast<const uint32_t *>(n_imgs > 1 ? img + (uint64_t)(f % n_imgs) * img_stride : img);      // (frame f reads image f mod n_imgs)
    uint32_t *__restrict__ o32 = reinterpret_cast<uint32_t *>(out + fd.out_off);
    const int cxl = tid & (kFwdTileW - 1);
    if (tx0 + cxl >= tx1) return;
    constexpr int kRowsPerStep = 4, kRowStride = kPwT / kFwdTileW;     // source reads of 4 tile rows in flight per lane
#pragma unroll 1
    for (int cy0 = tid / kFwdTileW; cy0 < ty1 - ty0; cy0 += kRowStride * kRowsPerStep) {
        uint32_t px[kRowsPerStep];
#pragma unroll
        for (int j = 0; j < kRowsPerStep; j++) {
            const int cyl = cy0 + kRowStride * j;
            px[j] = 0u;
            if (cyl >= ty1 - ty0) continue;
            const uint32_t w = s_win[cyl * kFwdTileW + cxl];
            if (w == 0u) continue;
            const int my = (int)((w - 1u) >> 16), mx = (int)((w - 1u) & 0xffffu);
            const int64_t sidx = (int64_t)(my + p.min_src_y) * W + (mx + p.min_src_x);                    // :960
            if (sidx >= 0 && sidx < (int64_t)W * H) px[j] = img32[sidx];
        }
#pragma unroll
        for (int j = 0; j < kRowsPerStep; j++) {
            const int cyl = cy0 + kRowStride * j;
            if (cyl < ty1 - ty0) o32[(size_t)(ty0 + cyl) * fd.obj_w + tx0 + cxl] = px[j];
        }
    }
}

void launch_fwd_geo(int kind, const double *d_mat, const uint8_t *img, int W, int H, const FrameDesc &fd, int32_t *win, uint8_t *out, hipStream_t stream)
{
    const int64_t n = (fd.obj_w > 0 && fd.obj_h > 0) ? (int64_t)fd.obj_w * fd.obj_h : 0;
    if (n == 0) return;
    launch_fill_i32(win, (size_t)n, -1, stream);
    dim3 grid((W + 255) / 256, H);
    if (kind == 0) hipLaunchKernelGGL(k_fwd_scatter_geo<0>, grid, dim3(256), 0, stream, d_mat, W, H, fd, win);
    else           hipLaunchKernelGGL(k_fwd_scatter_geo<1>, grid, dim3(256), 0, stream, d_mat, W, H, fd, win);
    hipLaunchKernelGGL(k_fwd_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, win, img, W, H, 0, 0, 0, 1, fd, out);
}

void launch_fwd_tiles(int kind, const FwdBatch &batch, int n_frames, int max_w, int max_h,
                      const uint8_t *img, int n_imgs, uint64_t img_stride, int W, int H, uint8_t *out, hipStream_t stream)
{
    if (n_frames <= 0 || max_w <= 0 || max_h <= 0) return;
    const dim3 grid((max_w + kFwdTileW - 1) / kFwdTileW, (max_h + kFwdTileH - 1) / kFwdTileH, n_frames);
    if (batch.params) {
        if (kind == 0) hipLaunchKernelGGL((k_fwd_tiles<0, false>), grid, dim3(256), 0, stream, batch, img, n_imgs, img_stride, W, H, out);
        else           hipLaunchKernelGGL((k_fwd_tiles<1, false>), grid, dim3(256), 0, stream, batch, img, n_imgs, img_stride, W, H, out);
    } else {
        if (kind == 0) hipLaunchKernelGGL((k_fwd_tiles<0, true>), grid, dim3(256), 0, stream, batch, img, n_imgs, img_stride, W, H, out);
        else           hipLaunchKernelGGL((k_fwd_tiles<1, true>), grid, dim3(256), 0, stream, batch, img, n_imgs, img_stride, W, H, out);
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

void launch_fwd_pw_tiles(const FwdPwTiles &p, int n_frames, int max_w, int max_h, const uint8_t *img, int n_imgs, uint64_t img_stride, int W, int H, uint8_t *out, hipStream_t stream)
{
    if (n_frames <= 0 || max_w <= 0 || max_h <= 0) return;
    if (p.T > 0) hipLaunchKernelGGL(k_fwd_pw_bins, dim3((p.T + 3) / 4, n_frames), dim3(64, 4), 0, stream, p);
    hipLaunchKernelGGL(k_fwd_pw_tiles, dim3(p.tsx, p.tsy, n_frames), dim3(kPwT), 0, stream, p, img, n_imgs, img_stride, W, H, out);
}

void launch_fwd_pw(const int32_t *fmap, const float *fwd, const uint8_t *img, int W, int H, int min_src_x, int min_src_y, int map_w, int map_h,
                   const FrameDesc &fd, int32_t *win, uint8_t *out, hipStream_t stream)
{
    const int64_t n = (fd.obj_w > 0 && fd.obj_h > 0) ? (int64_t)fd.obj_w * fd.obj_h : 0;
    if (n == 0) return;
    launch_fill_i32(win, (size_t)n, -1, stream);
    if (map_w > 0 && map_h > 0)
        hipLaunchKernelGGL(k_fwd_scatter_pw, dim3((map_w + 255) / 256, map_h), dim3(256), 0, stream, fmap, fwd, min_src_x, min_src_y, map_w, map_h, fd, win);
    hipLaunchKernelGGL(k_fwd_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, win, img, W, H, 1, min_src_x, min_src_y, map_w > 0 ? map_w : 1, fd, out);
}


} // namespace hg
