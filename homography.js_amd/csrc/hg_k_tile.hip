// hg_k_tile.hip -- inverse piecewise-affine warp of SHEARED meshes: k_pw_tile (round 4)
// Hand-written HIP for gfx950 (MI355X / CDNA4), wave64; fp64 coordinate math with contraction off so that nearest-neighbour
// source selection is bit-identical to the reference's JS doubles.
// Citations are file:line into the reference's Homography.js (v1.8.0).  Design notes: DESIGN.md §4.
#include "hg_dev.h"

namespace hg {

// ------------------------------------------------------------------------------------------------ k_pw_tile
// Same contract as k_pw_patch<SELF> (k_tri_setup in front, the workgroup evaluates its own spans, flags what it cannot hold), another
// decomposition: a workgroup owns a TILE of 8 output rows x 2048 columns, and every gather instruction reads eight RUNS of 8 pixels that
// FOLLOW THE SOURCE ROW -- run j of an 8-column octet holds the pixels (c, (j + round(s c)) mod 8), s = -m1 / m3 of the triangle under
// the block's centre: along it the source row is constant, so a run touches one source line (two when it straddles) where an 8-pixel
// piece of an OUTPUT row touches up to 8 under shear 1 (k_pw_patch: ~19 lines per gather instruction, here ~8-12).  s only orders the
// work: every pixel of the 64 x 8 block is visited exactly once whatever its value, and each pixel is computed exactly as everywhere
// else.  The taller workgroup is affordable because it is NARROW: spans are kept only where they overlap the tile's columns,
// candidates only for triangles that reach the tile in rows AND columns.
// Timing experiments that led here: EXPERIMENTS.md R4.3, R4.7; the persistent, wave-specialised form of this kernel was built in round 6
// and is slower (R6.3: git show 5f4414b:homography.js_amd/csrc/hg_k_tile.hip).
#ifndef HG_TILE_PB
#define HG_TILE_PB 4
#endif
#ifndef HG_TILE_COLS
#define HG_TILE_COLS 2048
#endif
constexpr int kTileRows = 8, kTileCols = HG_TILE_COLS, kTileBlocks = kTileCols / 64, kTileCap = kTileRowSpanCap, kTileRecs = 128, kTileCands = 192,
              kTileBinSlots = 8, kTilePitch = 72, kTilePB = HG_TILE_PB,
              kTileSpanPitch = kTileCap + 8;         // words between the rows' span blocks: 8 mod 64, so the eight rows a wave looks up at once start 8 banks apart (96: 4-way conflicts)

template <bool HIB>
__global__ __launch_bounds__(256) void k_pw_tile(PwMesh mesh, PwFrames fr, RowLists rl, uint8_t *__restrict__ out, int groups_per_xcd, int col_tiles,
                                                 int tile_cols, int32_t *__restrict__ status_next)
{
    const int bid = blockIdx.x, xcd = bid & ((1 << fr.xcc_log2) - 1), bi = bid >> fr.xcc_log2;
    const int bg = bi / col_tiles, ct = bi - bg * col_tiles;                // (the column tiles of a row group are neighbours in the grid)
    int f, gi;
    if (!frame_group(fr, xcd, bg, groups_per_xcd, f, gi)) return;           // (bands, rotating bands or dealt sub-bands: hg_dev.h)
    const int r0 = gi * kTileRows, t0 = ct * tile_cols;                     // (tile_cols: the frame width split evenly, a multiple of 64, <= kTileCols)
    const FrameDesc fd = fr.frames[f];
    if (bid == 0 && status_next) for (int i = threadIdx.x; i < fr.n_frames; i += 256) status_next[i] = 0;   // (see k_pw_rows)
    // the OTHER counter set, every row of the frame's block: clean for the next step (ping-pong, see k_pw_rows; the band counters live there)
    if (ct == 0 && (int)threadIdx.x < kTileRows && r0 + (int)threadIdx.x < rl.row_stride) rl.cnt_clear[(size_t)f * rl.row_stride + r0 + threadIdx.x] = 0;
    if (r0 >= fd.obj_h || fd.obj_w <= 0 || t0 >= fd.obj_w) return;

    __shared__ __align__(16) double s_rec[(kTileRecs + 1) * 6];               // {m0, m2, m4, m1, m3, m5} per candidate entry; last = NaN record
    __shared__ uint32_t s_lohi[kTileRows * kTileSpanPitch];                          // span cells [lo, hi) of the row, relative to the tile's first column
    __shared__ int s_key[kTileRows * kTileSpanPitch];                                // id << 14 | byte offset of the candidate's record
    __shared__ int s_bincnt[kTileBlocks * kTileRows];
    __shared__ __align__(4) uint8_t s_bin[kTileBlocks * kTileRows * kTileBinSlots];
    __shared__ int s_cand_tn[kTileCands], s_cand_y[kTileCands];
    __shared__ int s_rowcnt[kTileRows], s_ncand, s_fail;
    __shared__ uint32_t s_tile[4 * kTileRows * kTilePitch];
    __shared__ float s_slope[kTileBlocks];

    const int W = fd.obj_w;
    const int nrows = min(kTileRows, fd.obj_h - r0), ncols = min(tile_cols, W - t0), nblk = (ncols + 63) >> 6;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int i = threadIdx.x; i < kTileBlocks * kTileRows; i += 256) s_bincnt[i] = 0;
    if (threadIdx.x < kTileRows) s_rowcnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_ncand = 0; s_fail = 0; }
    if (threadIdx.x < 3) reinterpret_cast<double2 *>(s_rec + kTileRecs * 6)[threadIdx.x] = make_double2(NAN, NAN);
    __syncthreads();

    // ---- (1) candidates: triangles one of whose rows can write into rows r0 .. r0 + nrows - 1 (see k_pw_rows<SELF>) AND whose cells can
    // lie in columns [t0, t0 + ncols): a row's cells are (y - yOff) W + round(x), x between the vertex x's (+-1), so their columns are
    // [xlo, xhi] modulo W -- one interval, or two when it wraps over the row end (x-offset quirk, Appendix A-Q4).  Generous, never missing.
    const int T = mesh.n_tris;
    const int g_lo = r0 + fd.y_off, g_hi = r0 + nrows - 1 + fd.y_off;
    int n_src = T;
    const int4 *__restrict__ bent = nullptr;
    if (fr.band_ent) {
        const int bandi = r0 >> fr.band_rows_log2;          // (8-row tiles never straddle a 64-row band)
        n_src = min(fr.band_cnt[(size_t)f * fr.band_stride + bandi], fr.band_cap);
        bent = fr.band_ent + ((size_t)f * fr.n_bands + bandi) * fr.band_cap * 2;
    }
    const TriRange *__restrict__ trir = fr.trir + (size_t)f * T;
    const int2 *__restrict__ trix = fr.trix + (size_t)f * T;
    for (int i0 = 0; i0 < n_src; i0 += 256) {
        const int i = i0 + (int)threadIdx.x;
        int t = i, xlo = 0, xhi = -1;
        TriRange tr = TriRange{0, 0, 0, 0};
        if (i < n_src) {
            if (bent) {
                const int4 e = bent[2 * i], x = bent[2 * i + 1];
                t = e.x; tr.y_min = e.y; tr.y_end = e.z; tr.a = (int16_t)(e.w & 0xffff); tr.b = e.w >> 16; xlo = x.x; xhi = x.y;
            } else { tr = trir[i]; const int2 x = trix[i]; xlo = x.x; xhi = x.y; }
        }
        bool cols = false;
        if (xhi >= xlo) {
            if (xhi - xlo + 1 >= W) cols = true;
            else {
                int cl = xlo % W; if (cl < 0) cl += W;       // column of xlo
                const int ch = cl + (xhi - xlo);             // < 2 W
                cols = (cl < t0 + ncols && ch >= t0) || (cl - W < t0 + ncols && ch - W >= t0);
            }
        }
        int ylo0 = max(g_lo - tr.a, tr.y_min), n0 = min(g_hi - tr.b, tr.y_end - 1) - ylo0 + 1;
        int ylo1 = max(g_lo - tr.a - fd.obj_h, tr.y_min), n1 = min(g_hi - tr.b - fd.obj_h, tr.y_end - 1) - ylo1 + 1;
        if (!cols) { n0 = 0; n1 = 0; }
        const unsigned long long m0 = __ballot(n0 > 0), m1 = __ballot(n1 > 0);
        if ((m0 | m1) == 0ull) continue;                    // (wave-uniform)
        const unsigned long long m0b = __ballot(n0 > kTileRows), m1b = __ballot(n1 > kTileRows);
        const int c0 = __popcll(m0), c0b = __popcll(m0b), c1 = __popcll(m1), c1b = __popcll(m1b);
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_ncand, c0 + c0b + c1 + c1b);
        base = __builtin_amdgcn_readfirstlane(base);
        auto below = [&](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
        auto file = [&](int at, int y0, int n) { if (at < kTileCands) { s_cand_tn[at] = t | (min(n, 0xffff) << 16); s_cand_y[at] = y0; } };
        if (n0 > 0) file(base + below(m0), ylo0, min(n0, kTileRows));
        if (n0 > kTileRows) file(base + c0 + below(m0b), ylo0 + kTileRows, n0 - kTileRows);
        if (m1) {
            if (n1 > 0) file(base + c0 + c0b + below(m1), ylo1, min(n1, kTileRows));
            if (n1 > kTileRows) file(base + c0 + c0b + c1 + below(m1b), ylo1 + kTileRows, n1 - kTileRows);
        }
    }
    __syncthreads();
    const int nc = s_ncand;
    if (nc > kTileRecs) { if (threadIdx.x == 0) s_fail = 1 | (nc << 8); }
    // ---- (2) spans: 8 lanes per candidate entry, one source row each (predictXLimits :1172-1197 + the fill() indices :1124, the lean
    // form of span_cells: see k_pw_rows<SELF>); pieces cut at output-row boundaries, then to the tile's columns
    const float *__restrict__ ginv = fr.inv + (size_t)f * T * kInvStride;
    const double flen = (double)((int64_t)W * fd.obj_h), fW = (double)W;
    const Seg *__restrict__ gseg = fr.segs + (size_t)f * T * 3;
    if (nc <= kTileRecs) for (int c0 = 0; c0 < nc; c0 += 32) {
        const int c = c0 + ((int)threadIdx.x >> 3), jj = threadIdx.x & 7;
        if (c >= nc) continue;
        const int tn = s_cand_tn[c], t = tn & 0xffff, n = (int)((uint32_t)tn >> 16), ylo = s_cand_y[c];
        if (n >= 0xffff) { s_fail = 1; continue; }
        if (jj == 0) {
            const float4 ma = *reinterpret_cast<const float4 *>(ginv + (size_t)t * kInvStride);
            const float2 mb = *reinterpret_cast<const float2 *>(ginv + (size_t)t * kInvStride + 4);
            double2 *mrec = reinterpret_cast<double2 *>(s_rec + c * 6);
            mrec[0] = make_double2((double)ma.x, (double)ma.z);    // m0, m2
            mrec[1] = make_double2((double)mb.x, (double)ma.y);    // m4, m1
            mrec[2] = make_double2((double)ma.w, (double)mb.y);    // m3, m5
        }
        const Seg *__restrict__ sg = gseg + (size_t)t * 3;
        for (int j = jj; j < n; j += 8) {
            const int ys = ylo + j;
            const double y = (double)ys;
            double mn = INFINITY, mx = -INFINITY;
            const Seg q0 = sg[0], q1 = sg[1], q2 = sg[2];
            auto edge = [&](const Seg &q) {
                const double x = q.m == INFINITY ? q.b : (y - q.b) / q.m;
                const bool use = (y >= q.minY) & (y <= q.maxY) & !(q.m == 0.0);
                mn = (use & (x < mn)) ? x : mn;
                mx = (use & (x > mx)) ? x : mx;
            };
            edge(q0); edge(q1); edge(q2);
            const double base = (y - (double)fd.y_off) * fW;
            double rk = floor(mn); rk += (mn - rk >= 0.5) ? 1.0 : 0.0;
            double rf = floor(mx); rf += (mx - rf >= 0.5) ? 1.0 : 0.0;
            double vk = trunc(base + rk), vf = trunc(base + rf);
            vk = vk < 0.0 ? flen + vk : vk; vf = vf < 0.0 ? flen + vf : vf;
            const int k = (int)fmin(fmax(vk, 0.0), flen), fin = (int)fmin(fmax(vf, 0.0), flen);
            if (k >= fin) continue;
            int r = ys - fd.y_off;
            if (r < 0) r += fd.obj_h;
            if ((unsigned)r >= (unsigned)fd.obj_h || (unsigned)(k - r * W) >= (unsigned)W) r = k / W;
            if (r < r0) r = r0;
            for (; r < r0 + nrows; r++) {
                const int rb = r * W;
                if (rb >= fin) break;
                const int lo = max(max(k - rb, 0), t0) - t0, hi = min(min(fin - rb, W), t0 + ncols) - t0;      // ... and to the tile's columns
                if (lo >= hi) continue;
                const int row = r - r0;
                const int slot = atomicAdd(&s_rowcnt[row], 1);
                if (slot >= kTileCap) continue;             // (counted: the check below fails the tile)
                s_lohi[row * kTileSpanPitch + slot] = (uint32_t)lo | ((uint32_t)hi << 16);
                s_key[row * kTileSpanPitch + slot] = (t << kKeyShift) | (c * 48);
                // ... and into the bin of every 64-pixel block of the tile it overlaps, right here: a pass of its own over the span blocks
                // (one more barrier, every workgroup's full latency) cost 12-35 us of the kernel, EXPERIMENTS.md R4.7
                for (int b = lo >> 6; b <= (hi - 1) >> 6; b++) {
                    const int pos = atomicAdd(&s_bincnt[b * kTileRows + row], 1);      // (a count beyond the slots marks the bin as overfull)
                    if (pos < kTileBinSlots) s_bin[(b * kTileRows + row) * kTileBinSlots + pos] = (uint8_t)slot;
                }
            }
        }
    }
    __syncthreads();
    // ---- (3) limits: every thread reads the eight row counts (no further barrier)
    int fail = s_fail;
#pragma unroll
    for (int r = 0; r < kTileRows; r++) { const int cnt = s_rowcnt[r]; if (cnt > kTileCap && fail == 0) fail = 2 | (cnt << 8); }
    if (fail) {                                             // the host redoes the frame through the materialised map
        if (threadIdx.x == 0) flag_frame(fr, f, FRAME_LDS_OVERFLOW | (fail << 4));
        return;
    }

    const int c = lane & 7, q = lane >> 3;
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(frame_img(mesh, f)), 4, mesh.W * mesh.H, 0x00020000);   // (records of 4 bytes: gathers by pixel index, hg_struct_load_u32)
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(out + fd.out_off + (int64_t)r0 * W * 4, 0, nrows * W * 4, 0x00020000);
    const double bx_lo = (double)mesh.min_src_x + 0.5, bx_hi = (double)mesh.W + (double)mesh.min_src_x + 0.5;
    const double by_lo = (double)mesh.min_src_y + 0.5, by_hi = (double)mesh.H + (double)mesh.min_src_y + 0.5;
    const HiBounds hb = make_hi_bounds(bx_lo, bx_hi, by_lo, by_hi);
    const int nan_key = (int)0x80000000u | (kTileRecs * 48);
    uint32_t *tile = s_tile + wave * (kTileRows * kTilePitch);

    // One 64-column block, all 8 rows, in ONE pass: lane (q, c) holds the eight pixels (cb + 8 k + c, row), row = (q + sk) mod 8 -- the same
    // row for all eight (the run restarts in every octet), so one bin lookup serves them.  Gathers issued, not waited for.
    auto resolve_gather = [&](int blk, int sk, uint32_t px[8]) {
        const int cb = blk << 6;
        const int row = (q + sk) & (kTileRows - 1);
        int best[8];
#pragma unroll
        for (int k = 0; k < 8; k++) best[k] = nan_key;
        {   // the largest key over the spans of (row, block) that cover each pixel: larger id wins (== last writer of :852-858)
            const int bidx = blk * kTileRows + row;
            const int nb = s_bincnt[bidx];
            const uint8_t *bin = s_bin + bidx * kTileBinSlots;
            const int row_base = row * kTileSpanPitch;
            if (!__any(nb > kTileBinSlots)) {
                for (int p = 0; __any(p < nb); p++) {
                    int lo = 0, len = 0, key = 0;
                    if (p < nb) {
                        const int e = row_base + bin[p];
                        const uint32_t lh = s_lohi[e];
                        lo = (int)(lh & 0xffffu); len = (int)(lh >> 16) - lo; key = s_key[e];
                    }
                    const int d = cb + c - lo;
                    span_max4d(best, d, d + 8, d + 16, d + 24, len, key);
                    span_max4d(best + 4, d + 32, d + 40, d + 48, d + 56, len, key);
                }
            } else {                                        // an overfull bin: the row's whole list (slow, exact, rare)
                const int my_cnt = s_rowcnt[row];
                for (int i = 0; __any(i < my_cnt); i++) {
                    int lo = 0, len = 0, key = 0;
                    if (i < my_cnt) {
                        const uint32_t lh = s_lohi[row_base + i];
                        lo = (int)(lh & 0xffffu); len = (int)(lh >> 16) - lo; key = s_key[row_base + i];
                    }
                    const int d = cb + c - lo;
                    span_max4d(best, d, d + 8, d + 16, d + 24, len, key);
                    span_max4d(best + 4, d + 32, d + 40, d + 48, d + 56, len, key);
                }
            }
        }
        const double y = (double)(r0 + row + fd.y_off);
        const double xd0 = (double)(t0 + cb + c + fd.x_off);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            double h[8], rd[8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double2 *mrec = reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(s_rec) + (best[4 * half + k] & kKeyOffMask));
                const double2 m02 = mrec[0], m41 = mrec[1], m35 = mrec[2];
                const double xd = xd0 + (double)(8 * (4 * half + k));          // (integers far below 2^53: the fp64 add is exact; one instruction instead of an integer add and a conversion)
                // :1383-1384  (m0*x) + (m2*y) + m4: m2*y rounded on its own, m0*x exact in fp64 (see k_pw_rows)
                h[2 * k]     = fma(m02.x, xd, m02.y * y) + m41.x;
                h[2 * k + 1] = fma(m41.y, xd, m35.x * y) + m35.y;
            }
            round_x8(h, rd);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool inb = HIB ? hi_inb(hb, h[2 * k], h[2 * k + 1])
                                     : (bool)((int)(h[2 * k] >= bx_lo) & (int)(h[2 * k] < bx_hi) & (int)(h[2 * k + 1] >= by_lo) & (int)(h[2 * k + 1] < by_hi));   // :1047 (NaN fails)
                const int o = __mul24((int)dlo(rd[2 * k + 1]), mesh.W) + (int)dlo(rd[2 * k]);                               // :1048-1049, in pixels
                px[4 * half + k] = hg_struct_load_u32(src, inb ? o : -1, 0, 0, 0);
            }
        }
    };
    // 64 x 8 transpose through this wave's LDS tile (wave-synchronous), then 8 stores of 256 contiguous bytes each
    auto transpose_store = [&](int blk, int sk, const uint32_t px[8]) {
        const int row = (q + sk) & (kTileRows - 1);
#pragma unroll
        for (int k = 0; k < 8; k++) tile[row * kTilePitch + 8 * k + c] = px[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int xs = t0 + (blk << 6) + lane;
#pragma unroll
        for (int rw = 0; rw < kTileRows; rw++) {
            const uint32_t v = tile[rw * kTilePitch + lane];
            __builtin_amdgcn_raw_buffer_store_b32(v, dst, (xs < W && rw < nrows) ? (uint32_t)(rw * W + xs) * 4u : 0xffffffffu, 0, kStoreNT);
        }
        __builtin_amdgcn_wave_barrier();
    };
    // The run direction of a block: along the source row of the triangle under its centre (row 4, column 32): dy = -(m1 / m3) dx.  An
    // ordering choice only -- f32 arithmetic, clamped to one row per column.
    auto block_slope = [&](int blk) -> float {
        int best = nan_key;
        const int cb = blk << 6;
        const int bidx = blk * kTileRows + (kTileRows / 2);
        const int nb = min(s_bincnt[bidx], kTileBinSlots);
        for (int p = 0; p < nb; p++) {
            const int e = (kTileRows / 2) * kTileSpanPitch + s_bin[bidx * kTileBinSlots + p];
            const uint32_t lh = s_lohi[e];
            const int lo = (int)(lh & 0xffffu), len = (int)(lh >> 16) - lo;
            if ((unsigned)(cb + 32 - lo) < (unsigned)len) best = max(best, s_key[e]);
        }
        const double2 *mrec = reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(s_rec) + (best & kKeyOffMask));
        const float m1 = (float)mrec[1].y, m3 = (float)mrec[2].x;
        float sl = -m1 / m3;
        if (!(sl == sl) || best < 0) sl = 0.f;
        return fminf(fmaxf(sl, -1.f), 1.f);
    };
    // ... evaluated ONCE per tile: lane l of a wave takes that wave's l-th block (wave + 4 l) and leaves the slope in LDS.  (Every lane used
    // to evaluate every block's slope: ~40 vector instructions per 8 pixels for a number that is the same in all 64 lanes, and with a
    // shared source this kernel runs on its vector port -- EXPERIMENTS.md R6.4.)
    static_assert(kTileBlocks <= 4 * 64, "one lane per block of the wave");
    if (wave + 4 * lane < nblk) s_slope[wave + 4 * lane] = block_slope(wave + 4 * lane);      // (kept in LDS: a register held across the block loop cost a wave per SIMD)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    auto block_skew = [&](int blk) -> int { return (int)rintf(s_slope[blk] * (float)c); };
    // this wave's blocks: wave, wave + 4, ...; kTilePB of them in flight before the first is transposed and stored
    for (int blk = wave; blk < nblk; blk += 4 * kTilePB) {
        uint32_t px[kTilePB][8];
        int sk[kTilePB];
#pragma unroll
        for (int b = 0; b < kTilePB; b++) if (blk + 4 * b < nblk) { sk[b] = block_skew(blk + 4 * b); resolve_gather(blk + 4 * b, sk[b], px[b]); }
#pragma unroll
        for (int b = 0; b < kTilePB; b++) if (blk + 4 * b < nblk) transpose_store(blk + 4 * b, sk[b], px[b]);
    }
}

int launch_pw_tile(const PwMesh &mesh, const PwFrames &fr, const RowLists &rl, uint8_t *out, int max_obj_w, int32_t *status_next, hipStream_t stream)
{
    if (fr.n_frames <= 0 || fr.max_obj_h <= 0 || max_obj_w <= 0) return 0;
    const int nx = 1 << fr.xcc_log2;
    const int gpx = ((fr.max_obj_h + kTileRows - 1) / kTileRows + nx - 1) / nx;
    const int cts = (max_obj_w + kTileCols - 1) / kTileCols;
    const int tile_cols = (((max_obj_w + cts - 1) / cts) + 63) & ~63;          // the width split evenly over the column tiles (a 2170-pixel row: 2 x 1088, not 2048 + 122)
    PwFrames frs = fr;
    frs.sub_groups = sub_groups_of(fr, gpx);
    const dim3 grid((unsigned)padded_groups(gpx, frs.sub_groups) * (unsigned)nx * (unsigned)cts * (unsigned)fr.n_frames);
    const bool hib = !fr.no_hi_bounds && hi_bounds_ok(mesh.min_src_x, (int64_t)mesh.W + mesh.min_src_x, mesh.min_src_y, (int64_t)mesh.H + mesh.min_src_y);
    if (hib) hipLaunchKernelGGL((k_pw_tile<true>), grid, dim3(256), 0, stream, mesh, frs, rl, out, gpx, cts, tile_cols, status_next);
    else     hipLaunchKernelGGL((k_pw_tile<false>), grid, dim3(256), 0, stream, mesh, frs, rl, out, gpx, cts, tile_cols, status_next);
    return 500000 + kTilePB * 1000 + (hib ? 11 : 1);         // (variant code: see launch_pw_rows)
}

} // namespace hg
