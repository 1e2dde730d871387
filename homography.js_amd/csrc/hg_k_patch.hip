// hg_k_patch.hip -- inverse piecewise-affine warp of dense / sheared meshes: k_pw_patch
// Hand-written HIP for gfx950 (MI355X / CDNA4), wave64; fp64 coordinate math with contraction off so that nearest-neighbour
// source selection is bit-identical to the reference's JS doubles.
// Citations are file:line into the reference's Homography.js (v1.8.0).  Design notes: DESIGN.md §4.
#include "hg_dev.h"

namespace hg {

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
constexpr int kPatchRows = 4, kPatchCap = 200, kPatchRecs = 208, kPatchBins = 128, kPatchBinSlots = 8, kPatchHash = 1024, kPatchTilePitch = 80;
// (sized so that six workgroups fit a CU's 160 KB of LDS: 26.9 KB each)
static_assert(kPatchHash * 4 <= kPatchRows * kPatchBins * kPatchBinSlots, "the hash table lives in the bin-slot area");
// GLOBALREC variant for very dense meshes (up to 511 spans per row: README-scale, ~23 000 triangles on 4K): no matrix
// records in LDS at all -- a pixel reads its triangle's inverse matrix (6 floats, the tap array k_tri_spans fills, L2
// resident) from global memory and widens it itself; the LDS then holds 4 x 512 spans (30.4 KB, five workgroups per CU).
constexpr int kPatchCapDense = 512;
// SELF: one record per CANDIDATE triangle of the group -- the triangles whose row range reaches it, a slightly larger set than the
// triangles with a span in it (a triangle's last partial row, chunked entries at the window borders); 30.8 KB, five workgroups per CU,
// which is what the 86 registers of the 8-blocks-per-phase main loop allow anyway
constexpr int kPatchRecsSelf = 288;

// SELF (round 4): no row lists behind this kernel -- k_tri_setup ran, and the workgroup evaluates the spans of its four rows itself, like
// k_pw_rows<SELF> (hg_k_piecewise.hip): candidates from the frame's per-triangle row reach (scanned directly, or -- large meshes --
// from the entries k_tri_setup filed under this band of rows), one matrix record per CANDIDATE (no hash table: a candidate is a
// triangle), 4 lanes per candidate evaluating predictXLimits + the fill() indices of one source row each.
template <bool GLOBALREC, bool HIB, int PB = 1, bool SELF = false>      // PB: column blocks whose gathers are in flight before the first is stored
__global__ __launch_bounds__(256) void k_pw_patch(PwMesh mesh, PwFrames fr, RowLists rl, uint8_t *__restrict__ out,
                                                  int groups_per_xcd, int32_t *__restrict__ status_next)
{
    constexpr int CAPR = GLOBALREC ? kPatchCapDense : kPatchCap;            // spans per row
    using slot_t = typename std::conditional<GLOBALREC, uint16_t, uint8_t>::type;
    const int bid = blockIdx.x, xcd = bid & ((1 << fr.xcc_log2) - 1), bi = bid >> fr.xcc_log2;
    int f, gi;
    if (!frame_group(fr, xcd, bi, groups_per_xcd, f, gi)) return;           // (bands, rotating bands or dealt sub-bands: hg_dev.h)
    const int r0 = gi * kPatchRows;
    const FrameDesc fd = fr.frames[f];
    if (bid == 0 && status_next) for (int i = threadIdx.x; i < fr.n_frames; i += 256) status_next[i] = 0;   // (see k_pw_rows)
    // the OTHER counter set, every row of the frame's block: clean for the next step's k_tri_spans (ping-pong, see k_pw_rows)
    if ((int)threadIdx.x < kPatchRows && r0 + (int)threadIdx.x < rl.row_stride) rl.cnt_clear[(size_t)f * rl.row_stride + r0 + threadIdx.x] = 0;
    if (r0 >= fd.obj_h || fd.obj_w <= 0) return;

    constexpr int RECS = SELF ? kPatchRecsSelf : kPatchRecs;
    __shared__ __align__(16) double s_rec[GLOBALREC ? 6 : (RECS + 1) * 6];   // {m0, m2, m4, m1, m3, m5} per triangle; last = NaN record
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
    // (the source window, also for the prologue's span flags)
    const double bx_lo0 = (double)mesh.min_src_x + 0.5, bx_hi0 = (double)mesh.W + (double)mesh.min_src_x + 0.5;
    const double by_lo0 = (double)mesh.min_src_y + 0.5, by_hi0 = (double)mesh.H + (double)mesh.min_src_y + 0.5;
    const HiBounds hb0 = make_hi_bounds(bx_lo0, bx_hi0, by_lo0, by_hi0);
    const bool flag_spans = SELF && fr.safe_spans_patch != 0;       // (uniform; host option)
    const float *__restrict__ ginv0 = fr.inv + (size_t)f * mesh.n_tris * kInvStride;     // this frame's inverse matrices (tap array)
    int cnts[kPatchRows], cmax = 0;
    bool bad1;
    if constexpr (SELF) {
        static_assert(!GLOBALREC, "the self-span prologue keeps one record per candidate in LDS");
        __shared__ int s_ncand, s_rowcnt[kPatchRows];
        int *s_cand_tn = reinterpret_cast<int *>(s_tile), *s_cand_y = s_cand_tn + RECS;      // (the transpose tiles are idle until the main loop)
        static_assert(2 * RECS <= 4 * kPatchRows * kPatchTilePitch, "candidate list fits the tile area");
        for (int i = threadIdx.x; i < kPatchRows * kPatchBins; i += 256) s_bincnt[i] = 0;
        if (threadIdx.x < kPatchRows) s_rowcnt[threadIdx.x] = 0;
        if (threadIdx.x == 0) { s_ncand = 0; s_fail = nbins > kPatchBins ? 1 : 0; }
        if (threadIdx.x < 3) reinterpret_cast<double2 *>(s_rec + RECS * 6)[threadIdx.x] = make_double2(NAN, NAN);
        __syncthreads();
        // (1) candidates: can a row of the triangle write into rows r0 .. r0 + nrows - 1?  (see k_pw_rows<SELF>; int32 throughout)
        const int lane_ = threadIdx.x & 63;
        const int g_lo = r0 + fd.y_off, g_hi = r0 + nrows - 1 + fd.y_off;
        const int T = mesh.n_tris;
        int n_src = T;
        const int4 *__restrict__ bent = nullptr;
        if (fr.band_ent) {
            const int band = r0 >> fr.band_rows_log2;
            n_src = min(fr.band_cnt[(size_t)f * fr.band_stride + band], fr.band_cap);      // (an overfull band flagged the frame in k_tri_setup)
            bent = fr.band_ent + ((size_t)f * fr.n_bands + band) * fr.band_cap * 2;        // (two int4 per entry; the second holds the column reach k_pw_tile uses)
        }
        const TriRange *__restrict__ trir = fr.trir + (size_t)f * T;
        for (int i0 = 0; i0 < n_src; i0 += 256) {
            const int i = i0 + (int)threadIdx.x;
            int t = i; TriRange tr = TriRange{0, 0, 0, 0};
            if (i < n_src) {
                if (bent) { const int4 e = bent[2 * i]; t = e.x; tr.y_min = e.y; tr.y_end = e.z; tr.a = (int16_t)(e.w & 0xffff); tr.b = e.w >> 16; }
                else tr = trir[i];
            }
            const int ylo0 = max(g_lo - tr.a, tr.y_min), n0 = min(g_hi - tr.b, tr.y_end - 1) - ylo0 + 1;
            const int ylo1 = max(g_lo - tr.a - fd.obj_h, tr.y_min), n1 = min(g_hi - tr.b - fd.obj_h, tr.y_end - 1) - ylo1 + 1;
            const unsigned long long m0 = __ballot(n0 > 0), m1 = __ballot(n1 > 0);
            if ((m0 | m1) == 0ull) continue;                // (wave-uniform)
            const unsigned long long m0b = __ballot(n0 > 4), m1b = __ballot(n1 > 4);
            const int c0 = __popcll(m0), c0b = __popcll(m0b), c1 = __popcll(m1), c1b = __popcll(m1b);
            int base = 0;
            if (lane_ == 0) base = atomicAdd(&s_ncand, c0 + c0b + c1 + c1b);
            base = __builtin_amdgcn_readfirstlane(base);
            auto below = [&](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
            auto file = [&](int at, int y0, int n) { if (at < RECS) { s_cand_tn[at] = t | (min(n, 0xffff) << 16); s_cand_y[at] = y0; } };
            if (n0 > 0) file(base + below(m0), ylo0, min(n0, 4));
            if (n0 > 4) file(base + c0 + below(m0b), ylo0 + 4, n0 - 4);
            if (m1) {
                if (n1 > 0) file(base + c0 + c0b + below(m1), ylo1, min(n1, 4));
                if (n1 > 4) file(base + c0 + c0b + c1 + below(m1b), ylo1 + 4, n1 - 4);
            }
        }
        __syncthreads();
        const int nc = s_ncand;
        if (nc > RECS) { if (threadIdx.x == 0) s_fail = 1 | (nc << 8); }
        // (2) spans: 4 lanes per candidate entry, one source row each; lane 0 of the entry also writes the candidate's matrix record
        const double flen = (double)((int64_t)W * fd.obj_h), fW = (double)W;      // (len < 2^31: fill_frames)
        const Seg *__restrict__ gseg = fr.segs + (size_t)f * T * 3;
        if (nc <= RECS) for (int c0 = 0; c0 < nc; c0 += 64) {
            const int c = c0 + ((int)threadIdx.x >> 2), jj = threadIdx.x & 3;
            if (c >= nc) continue;
            const int tn = s_cand_tn[c], t = tn & 0xffff, n = (int)((uint32_t)tn >> 16), ylo = s_cand_y[c];
            if (n >= 0xffff) { s_fail = 1; continue; }      // (absurd reach: the map path takes the frame)
            const float4 ma = *reinterpret_cast<const float4 *>(ginv0 + (size_t)t * kInvStride);
            const float2 mb = *reinterpret_cast<const float2 *>(ginv0 + (size_t)t * kInvStride + 4);
            if (jj == 0) {
                double2 *mrec = reinterpret_cast<double2 *>(s_rec + c * 6);
                mrec[0] = make_double2((double)ma.x, (double)ma.z);    // m0, m2
                mrec[1] = make_double2((double)mb.x, (double)ma.y);    // m4, m1
                mrec[2] = make_double2((double)ma.w, (double)mb.y);    // m3, m5
            }
            const Seg *__restrict__ sg = gseg + (size_t)t * 3;
            for (int j = jj; j < n; j += 4) {
                const int ys = ylo + j;
                const double y = (double)ys;
                double mn = INFINITY, mx = -INFINITY;       // predictXLimits :1172-1197 (the lean form of span_cells, see k_pw_rows<SELF>)
                auto edge = [&](const Seg &q) {
                    const double x = q.m == INFINITY ? q.b : (y - q.b) / q.m;
                    const bool use = (y >= q.minY) & (y <= q.maxY) & !(q.m == 0.0);
                    mn = (use & (x < mn)) ? x : mn;
                    mx = (use & (x > mx)) ? x : mx;
                };
                if constexpr (PB >= 8) {                    // the 8-blocks-per-phase main loop holds 86 registers anyway: all three edge
                    const Seg q0 = sg[0], q1 = sg[1], q2 = sg[2];       // equations in one round trip
                    edge(q0); edge(q1); edge(q2);
                } else {
#pragma unroll 1
                    for (int e = 0; e < 3; e++) edge(sg[e]);
                }
                const double base = (y - (double)fd.y_off) * fW;               // :1124 under TypedArray.fill's index rules
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
                    const int lo = max(k - rb, 0), hi = min(fin - rb, W);
                    if (lo >= hi) continue;
                    const int row = r - r0;
                    const int slot = atomicAdd(&s_rowcnt[row], 1);
                    if (slot >= CAPR - 1) continue;         // (counted: the check below fails the group)
                    s_lohi[row * CAPR + slot] = (uint32_t)lo | ((uint32_t)hi << 16);
                    // bit 0 "unsafe" (record offsets are multiples of 48): clear only when BOTH end pixels of the piece pass the source bounds
                    // test :1047, evaluated exactly as the pixel body evaluates them -- then every pixel between them passes (k_pw_rows)
                    int unsafe = 1;
                    if (flag_spans) {
                        const double yr = (double)(r + fd.y_off), xa = (double)(lo + fd.x_off), xb = (double)(hi - 1 + fd.x_off);
                        const double cy0 = (double)ma.z * yr, cy1 = (double)ma.w * yr;
                        double h[4] = { fma((double)ma.x, xa, cy0) + (double)mb.x, fma((double)ma.y, xa, cy1) + (double)mb.y,
                                        fma((double)ma.x, xb, cy0) + (double)mb.x, fma((double)ma.y, xb, cy1) + (double)mb.y }, rd[4];
                        round_x4(h, rd);
                        const bool ia = HIB ? hi_inb(hb0, h[0], h[1]) : (bool)((int)(h[0] >= bx_lo0) & (int)(h[0] < bx_hi0) & (int)(h[1] >= by_lo0) & (int)(h[1] < by_hi0));
                        const bool ib = HIB ? hi_inb(hb0, h[2], h[3]) : (bool)((int)(h[2] >= bx_lo0) & (int)(h[2] < bx_hi0) & (int)(h[3] >= by_lo0) & (int)(h[3] < by_hi0));
                        unsafe = (ia && ib) ? 0 : 1;
                    }
                    s_key[row * CAPR + slot] = (t << kKeyShift) | (c * 48) | unsafe;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kPatchRows; j++) { cnts[j] = j < nrows ? s_rowcnt[j] : 0; cmax = max(cmax, cnts[j]); }
        bad1 = s_fail != 0 || cmax > CAPR - 1;
        if (bad1 && threadIdx.x == 0 && s_fail == 0) s_fail = 2 | (cmax << 8);
    } else {
    const int32_t *cntp = rl.cnt + (size_t)f * rl.row_stride + r0;
#pragma unroll
    for (int j = 0; j < kPatchRows; j++) { cnts[j] = j < nrows ? cntp[j] : 0; cmax = max(cmax, cnts[j]); }
    for (int i = threadIdx.x; i < kPatchRows * kPatchBins; i += 256) s_bincnt[i] = 0;
    if (!GLOBALREC) for (int i = threadIdx.x; i < kPatchHash; i += 256) s_hash[i] = 0u;
    if (threadIdx.x == 0) { s_nrec = 0; s_fail = (cmax > rl.cap || cmax > CAPR - 1 || nbins > kPatchBins) ? 1 : 0; }
    __syncthreads();
    const bool bad0 = s_fail != 0;

    // ---- phase 1: span lists -> LDS; each triangle of the group gets ONE matrix record (hash on the id: the thread that
    // claims the bucket writes the record and publishes its index; the others remember the bucket and read it later)
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
    if (!GLOBALREC && threadIdx.x < 3) reinterpret_cast<double2 *>(s_rec + RECS * 6)[threadIdx.x] = make_double2(NAN, NAN);
    __syncthreads();
    // ---- phase 2: keys (id << 14 | record offset), once every record index is published
    bad1 = s_fail != 0;
    n_mine = 0;
    if (!GLOBALREC && !bad1) for (int e = threadIdx.x; e < kPatchRows * CAPR; e += 256) {
        const int bucket = my_bucket[n_mine++];
        if (bucket >= 0) s_key[e] = (s_key[e] << kKeyShift) | (int)(((s_hash[bucket] & 0xffffu) - 1u) * 48u) | 1;
    }
    if (!GLOBALREC) __syncthreads();
    }
    // ---- phase 3: the hash table is dead, its memory becomes the bins: span index -> every 64-pixel column it overlaps
    if (!bad1) for (int e = threadIdx.x; e < kPatchRows * CAPR; e += 256) {
        const int rr = e / CAPR, i = e - rr * CAPR;
        const int cnt = cnts[0] * (rr == 0) + cnts[1] * (rr == 1) + cnts[2] * (rr == 2) + cnts[3] * (rr == 3);
        if (i < cnt) {
            const uint32_t lh = s_lohi[e];
            const int lo = (int)(lh & 0xffffu), hi = (int)(lh >> 16);
            for (int b = lo >> 6; b <= (hi - 1) >> 6 && b < nbins; b++) {
                const int pos = atomicAdd(&s_bincnt[b * kPatchRows + rr], 1);          // (a count beyond the slots marks the bin as overfull)
                if (pos < kPatchBinSlots) s_bin[(b * kPatchRows + rr) * kPatchBinSlots + pos] = (slot_t)i;
            }
        }
    }
    __syncthreads();
    if (s_fail) {                                           // the host redoes the frame through the materialised map
        // (SELF: bits 4.. say which limit -- 1 | candidates << 8, or 2 | longest row << 8 -- for whoever reads the status word in a debugger)
        if (threadIdx.x == 0) flag_frame(fr, f, FRAME_LDS_OVERFLOW | (SELF ? (s_fail << 4) : 0));
        return;
    }

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rr = lane >> 4;                               // lane = (row of the group, columns (lane & 15) + 16k of the block)
    int ck[4];
#pragma unroll
    for (int k = 0; k < 4; k++) ck[k] = (lane & 15) + 16 * k;
    const double y = (double)(r0 + rr + fd.y_off);
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(frame_img(mesh, f)), 4, mesh.W * mesh.H, 0x00020000);   // (records of 4 bytes: gathers by pixel index, hg_struct_load_u32)
    // output: the group's rows as one raw buffer; lanes of rows past the frame end and pixels past the row end get an
    // offset the hardware range check drops
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(out + fd.out_off + (int64_t)r0 * W * 4, 0, nrows * W * 4, 0x00020000);
    const double bx_lo = (double)mesh.min_src_x + 0.5, bx_hi = (double)mesh.W + (double)mesh.min_src_x + 0.5;
    const double by_lo = (double)mesh.min_src_y + 0.5, by_hi = (double)mesh.H + (double)mesh.min_src_y + 0.5;
    const HiBounds hb = make_hi_bounds(bx_lo, bx_hi, by_lo, by_hi);      // HIB: :1047 on the high dwords of h (hg_dev.h)
    const int nan_key = GLOBALREC ? -1 : ((int)0x80000000u | (RECS * 48) | 1);      // ("no triangle" is unsafe: its pixels must come out as offset 0xffffffff)
    const int row_base = rr * CAPR;
    const int my_cnt = cnts[0] * (rr == 0) + cnts[1] * (rr == 1) + cnts[2] * (rr == 2) + cnts[3] * (rr == 3);
    uint32_t *tile = s_tile + wave * (kPatchRows * kPatchTilePitch);
    const float *__restrict__ ginv = fr.inv + (size_t)f * mesh.n_tris * kInvStride;     // GLOBALREC: this frame's inverse matrices

    // One 64-pixel-wide column block, all 4 rows: span lookup, coordinates, gathers issued (not waited for).
    auto resolve_gather = [&](int cw, uint32_t px[4]) {
        const int c0 = cw << 6;                             // pixel k of the lane: (c0 + ck[k], r0 + rr)
        int best[4] = { nan_key, nan_key, nan_key, nan_key };
        const int bidx = cw * kPatchRows + rr;      // (bins of the four rows side by side: the four row groups of a wave read different banks)
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
                const double2 *mrec = reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(s_rec) + (best[k] & (kKeyOffMask & ~15)));
                const double2 m02 = mrec[0], m41 = mrec[1], m35 = mrec[2];
                m0 = m02.x; m2 = m02.y; m4 = m41.x; m1 = m41.y; m3 = m35.x; m5 = m35.y;
            }
            const double xd = (double)(c0 + ck[k] + fd.x_off);
            // :1383-1384  (m0*x) + (m2*y) + m4: m2*y rounded on its own, m0*x exact in fp64 (see k_pw_rows)
            h[2 * k]     = fma(m0, xd, m2 * y) + m4;
            h[2 * k + 1] = fma(m1, xd, m3 * y) + m5;
        }
        // every pixel of the block resolved to a span whose ends are inside the source window: no bounds test, Math.round with one add (k_pw_rows)
        if (flag_spans && __ballot(((best[0] | best[1] | best[2] | best[3]) & 1) != 0) == 0ull) {
            int r[8];
            round_half_x4(h, r); round_half_x4(h + 4, r + 4);
#pragma unroll
            for (int k = 0; k < 4; k++)
                px[k] = hg_struct_load_u32(src, __mul24(r[2 * k + 1], mesh.W) + r[2 * k], 0, 0, 0);
            return;
        }
        round_x8(h, rd);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool inb = HIB ? hi_inb(hb, h[2 * k], h[2 * k + 1])
                                 : (bool)((int)(h[2 * k] >= bx_lo) & (int)(h[2 * k] < bx_hi) & (int)(h[2 * k + 1] >= by_lo) & (int)(h[2 * k + 1] < by_hi));   // :1047 (NaN fails)
            const int o = __mul24((int)dlo(rd[2 * k + 1]), mesh.W) + (int)dlo(rd[2 * k]);                                   // :1048-1049, in pixels
            px[k] = hg_struct_load_u32(src, inb ? o : -1, 0, 0, 0);
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
    // (Measured and dropped, round 4: a fast path for blocks no span reaches -- a third of C4's -- written as zeros without lookup,
    //  transform, gather or transpose.  Per block, the branches make the compiler wait for each block's loads inside its branch: C3 with
    //  one source per frame 0.82 -> 1.03 ms; per phase of 8 contiguous blocks it costs 120 instead of 86 registers: C4 0.33 -> 0.365,
    //  C5 shared 0.423 -> 0.462.)
    // This wave's column blocks cw = wave, wave + 4, ..., PB of them per phase: all their gathers are in flight before the first
    // block is transposed and stored (loads and stores share vmcnt, see k_pw_rows).  Round 2 measured 2 blocks per phase on C5
    // with a shared source and found nothing (0.504 vs 0.503 ms); round 3, same box, blocks per phase 1 / 2 / 4 / 8 / 16:
    // C5 shared 0.419 / 0.412 / 0.405 / 0.392-0.397 / 0.404 ms, C5 one source per frame 0.612 / 0.593 / 0.588 / 0.565-0.586 / 0.557,
    // C3 one source per frame 0.924 / 0.871 / 0.847 / 0.819-0.823 / 0.828 -- 8 is the default (86 VGPRs: still the 5 workgroups
    // per CU the LDS allows; 16 takes 128).
    for (int cw = wave; cw < nbins; cw += 4 * PB) {
        uint32_t px[PB][4];
#pragma unroll
        for (int b = 0; b < PB; b++) if (cw + 4 * b < nbins) resolve_gather(cw + 4 * b, px[b]);
#pragma unroll
        for (int b = 0; b < PB; b++) if (cw + 4 * b < nbins) transpose_store(cw + 4 * b, px[b]);
    }
}

int launch_pw_patch(const PwMesh &mesh, const PwFrames &fr, const RowLists &rl, uint8_t *out, int32_t *status_next, bool global_records, hipStream_t stream)
{
    if (fr.n_frames <= 0 || fr.max_obj_h <= 0) return 0;
    int code = 0;                                            // (variant code: see launch_pw_rows)
    const int nx = 1 << fr.xcc_log2;
    const int gpx = ((fr.max_obj_h + kPatchRows - 1) / kPatchRows + nx - 1) / nx;
    PwFrames frs = fr;
    frs.sub_groups = sub_groups_of(fr, gpx);
    const dim3 grid((unsigned)padded_groups(gpx, frs.sub_groups) * (unsigned)nx * (unsigned)fr.n_frames);
    const bool hib = !fr.no_hi_bounds && hi_bounds_ok(mesh.min_src_x, (int64_t)mesh.W + mesh.min_src_x, mesh.min_src_y, (int64_t)mesh.H + mesh.min_src_y);
#define HG_PATCH(G, HB, PBV, SF) do { code = ((G) ? 800000 : 400000) + (PBV) * 1000 + ((HB) ? 10 : 0) + ((SF) ? 1 : 0); \
        hipLaunchKernelGGL((k_pw_patch<G, HB, PBV, SF>), grid, dim3(256), 0, stream, mesh, frs, rl, out, gpx, status_next); } while (0)
    // ONE depth: 8 column blocks per gather / store phase (the census of round 6 found the 1- / 2- / 4-block instantiations never picked
    // by the layout policy: deleted, EXPERIMENTS.md R6.6); the fp64-bounds form keeps its single block
    if (fr.self_spans && !global_records) {                  // own spans (k_tri_setup in front, no row lists)
        if (!hib) HG_PATCH(false, false, 1, true); else HG_PATCH(false, true, 8, true);
        return code;
    }
    if (global_records) { if (hib) HG_PATCH(true, true, 1, false); else HG_PATCH(true, false, 1, false); }
    else if (!hib) HG_PATCH(false, false, 1, false);
    else HG_PATCH(false, true, 8, false);
#undef HG_PATCH
    return code;
}

} // namespace hg
