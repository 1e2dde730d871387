// hg_api_piecewise.hip -- the C ABI, part 3: _inversePiecewiseAffineWarp: mesh, frame sets (staging, layout estimate), the fused
// path and its deferred redo, hg_sync, the taps.
#include "hg_ctx.h"

// ------------------------------------------------------------------------------------------------ piecewise affine
// NaN is a legal (if useless) coordinate -- the reference then simply draws nothing for that triangle -- but magnitudes beyond
// kMaxCoord (Infinity included) are refused: the row loops of the rasterisers are bounded under that assumption (hg_math.h).
static bool coords_ok(const float *p, size_t n)
{
    for (size_t i = 0; i < n; i++) if (std::fabs((double)p[i]) > kMaxCoord) return false;      // (NaN compares false)
    return true;
}

extern "C" int hg_piecewise_set_mesh(hg_ctx *c, const float *src, int n_pts, const uint32_t *tris, int n_tris, int msx, int msy)
{
    HG_TRY(bind(c));
    if (!src || n_pts <= 0 || n_tris < 0 || (!tris && n_tris > 0)) return fail(c, HG_ERR_INVALID, "hg_piecewise_set_mesh: bad arguments");
    if (!coords_ok(src, (size_t)n_pts * 2))
        return fail(c, HG_ERR_INVALID, "hg_piecewise_set_mesh: a source coordinate is infinite or beyond 2^24 in magnitude");
    // bindings re-send the mesh on every warp (the reference keeps it cached, :742, :758): an identical mesh keeps the
    // device copies and what was derived from them (the forward triangle map)
    if (c->have_mesh && n_pts == c->n_pts && n_tris == c->n_tris && msx == c->min_src_x && msy == c->min_src_y &&
        std::memcmp(src, c->h_src.data(), sizeof(float) * 2 * (size_t)n_pts) == 0 &&
        (n_tris == 0 || std::memcmp(tris, c->h_tris.data(), sizeof(uint32_t) * 3 * (size_t)n_tris) == 0))
        return HG_OK;
    HG_TRY(hg_sync(c));
    HG_TRY(ensure(c, c->d_src, c->src_cap, (size_t)n_pts * 2));
    HG_TRY(ensure(c, c->d_tris, c->tris_cap, (size_t)std::max(n_tris, 1) * 3));
    HIP_TRY(c, hipMemcpyAsync(c->d_src, src, sizeof(float) * 2 * n_pts, hipMemcpyHostToDevice, c->stream));
    if (n_tris > 0) HIP_TRY(c, hipMemcpyAsync(c->d_tris, tris, sizeof(uint32_t) * 3 * n_tris, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->n_pts = n_pts; c->n_tris = n_tris; c->min_src_x = msx; c->min_src_y = msy;
    c->h_tris.assign(tris, tris + (size_t)3 * n_tris);
    c->h_src.assign(src, src + (size_t)2 * n_pts);
    c->fmap_valid = false;
    c->have_mesh = true;
    c->mesh_gen++;
    c->fwd_pw_tiles_disabled = false; c->fwd_pw_cap = 64;     // (learned on the previous mesh)
    c->pw_self_disabled = false; c->pw_tile_disabled = false; c->pw_patch_disabled = false;
    c->pw_frames.clear(); c->pw_setup_done = false;
    return HG_OK;
}

// Largest number of triangles whose fillTriangle row range (:1113-1120) covers one output row, over the uploaded frames:
// an estimate of the longest per-row span list, used ONLY to pick k_pw_rows' layout (4 rows per workgroup with 64 LDS
// slots each, or 1 row with all 256); the kernel checks the real counts and is exact either way.
static int max_row_cover(const hg_ctx *c, const float *dst, double *mean_tri_rows, int *max_group_tris, double *mean_shear, int *max_tri_rows, double *fill)
{
    double tallest = 0.0, band_worst = 1.0;
    const int nb = 1 << c->xcc_log2;
    std::vector<double> band((size_t)nb, 0.0);
    int worst = 0, worst_group = 0;
    double rows_total = 0.0, tris_total = 0.0, shear_total = 0.0, shear_n = 0.0;
    std::vector<int> diff, tdiff, starts;
    const int shear_stride = std::max(1, c->n_tris / 1024);
    for (size_t f = 0; f < c->pw_frames.size(); f++) {
        const FrameDesc &fd = c->pw_frames[f];
        if (fd.obj_w <= 0 || fd.obj_h <= 0) continue;
        const float *dp = dst + f * (size_t)c->n_pts * 2;
        diff.assign((size_t)fd.obj_h + 2, 0);
        starts.assign((size_t)fd.obj_h + 2, 0);
        tdiff.assign((size_t)fd.obj_h + 2, 0);
        for (int t = 0; t < c->n_tris; t++) {
            double lo = INFINITY, hi = -INFINITY;
            bool ok = true;
            double sx[3], sy[3], dx[3], dy[3];
            for (int k = 0; k < 3; k++) {
                const uint32_t v = c->h_tris[3 * (size_t)t + k];
                if (v >= (uint32_t)c->n_pts) { ok = false; break; }
                const double y = dp[2 * (size_t)v + 1];
                if (!(y == y)) { ok = false; break; }
                lo = std::min(lo, y); hi = std::max(hi, y);
                sx[k] = c->h_src[2 * (size_t)v]; sy[k] = c->h_src[2 * (size_t)v + 1]; dx[k] = dp[2 * (size_t)v]; dy[k] = y;
            }
            if (!ok) continue;
            if (t % shear_stride == 0) {   // |d(source row) / d(output x)| of the triangle's inverse map (a sample is enough): how many source lines 64 consecutive output
                // pixels spread over (plain doubles: an estimate, never used for pixels)
                const double e1x = sx[1] - sx[0], e1y = sy[1] - sy[0], e2x = sx[2] - sx[0], e2y = sy[2] - sy[0];
                const double f1x = dx[1] - dx[0], f1y = dy[1] - dy[0], f2x = dx[2] - dx[0], f2y = dy[2] - dy[0];
                const double D = e1x * e2y - e2x * e1y;
                const double a = (f1x * e2y - f2x * e1y) / D, cc = (e1x * f2x - e2x * f1x) / D;
                const double b = (f1y * e2y - f2y * e1y) / D, d = (e1x * f2y - e2x * f1y) / D;
                const double sh = std::fabs(b / (a * d - b * cc));
                if (sh == sh && sh < 1e6) { shear_total += sh; shear_n += 1.0; }
            }
            // rows [trunc(minY), ceil(maxY)) - yOff, one more below for spans that spill over the row end (x-offset quirk)
            const double a = std::max(std::trunc(lo) - fd.y_off, 0.0), b = std::min(std::ceil(hi) - fd.y_off + 1.0, (double)fd.obj_h);
            if (!(a < b)) continue;
            diff[(size_t)a] += 1; diff[(size_t)b] -= 1;
            rows_total += b - a; tris_total += 1.0;
            tallest = std::max(tallest, std::min(std::ceil(hi) - std::trunc(lo), 1.0e6));       // rows of fillTriangle's loop :1113-1120

            // tighter, for the triangles-per-group estimate: a triangle has spans on the integer rows inside [minY, maxY]
            // (:1179; triangles that only touch a row at a vertex between two integers do not count), plus the spill row when
            // the window is offset in x
            const double ta = std::max(std::ceil(lo) - fd.y_off, 0.0), tb = std::min(std::floor(hi) - fd.y_off + (fd.x_off != 0 ? 2.0 : 1.0), (double)fd.obj_h);
            if (ta < tb) { tdiff[(size_t)ta] += 1; tdiff[(size_t)tb] -= 1; starts[(size_t)ta] += 1; }
        }
        int run = 0, trun = 0;
        int group = 0;                                       // triangles with spans in the 4-row group the row belongs to
        std::fill(band.begin(), band.end(), 0.0);
        for (int r = 0; r < fd.obj_h; r++) {
            run += diff[r];
            trun += tdiff[r];
            worst = std::max(worst, run);
            band[(size_t)((int64_t)r * nb / fd.obj_h)] += 2.0 + run;          // a row's cost ~ a constant + its spans (layout heuristic only)
            group = (r % kRowGroup == 0) ? trun : group + starts[r];
            worst_group = std::max(worst_group, group);
        }
        double bsum = 0.0, bmax = 0.0;
        for (double v : band) { bsum += v; bmax = std::max(bmax, v); }
        if (bsum > 0) band_worst = std::max(band_worst, bmax * nb / bsum);
    }
    *mean_tri_rows = tris_total > 0 ? rows_total / tris_total : 0.0;
    *max_group_tris = worst_group;
    *mean_shear = shear_n > 0 ? shear_total / shear_n : 0.0;
    *max_tri_rows = (int)tallest;
    *fill = band_worst;                                      // heaviest XCD band / mean band, over the frames (1 = even rows)
    return worst;
}

static int pw_set_frames_impl(hg_ctx *c, const float *dst, const hg_geom *geoms, const size_t *offs, int n);

// Transactional wrapper: if validation, an allocation or an upload fails part-way, the context is left WITHOUT a frame set
// (the next warp returns HG_ERR_STATE) rather than with n new host-side frames over device buffers sized for the old set.
extern "C" int hg_piecewise_set_frames(hg_ctx *c, const float *dst, const hg_geom *geoms, const size_t *offs, int n)
{
    const int rc = pw_set_frames_impl(c, dst, geoms, offs, n);
    if (rc != HG_OK && c) {
        const std::string why = c->err;
        (void)hg_sync(c);                                   // queued runs of the old set are settled against the old set
        c->pw_frames.clear(); c->pw_setup_done = false; c->rows_clean = false;
        c->err = why; g_err = why;
    }
    return rc;
}

static int pw_set_frames_impl(hg_ctx *c, const float *dst, const hg_geom *geoms, const size_t *offs, int n)
{
    HG_TRY(bind(c));
    if (!c->have_mesh) return fail(c, HG_ERR_STATE, "no mesh: call hg_piecewise_set_mesh first");
    if (!dst || !geoms || n <= 0) return fail(c, HG_ERR_INVALID, "hg_piecewise_set_frames: bad arguments");
    if (!coords_ok(dst, (size_t)n * c->n_pts * 2))
        return fail(c, HG_ERR_INVALID, "hg_piecewise_set_frames: a destiny coordinate is infinite or beyond 2^24 in magnitude "
                                       "(the reference's fillTriangle row loop would run for that many rows, forever for Infinity)");
    // Queued runs are NOT waited for: the uploads below are ordered behind them on the stream, and each of them keeps its own
    // staged copy of the set it warped (deferred redo, hg_sync).  Only a staging slot that a queued run still refers to forces a
    // settlement first (a caller that uploads 64 sets per run).
    const size_t T = (size_t)std::max(c->n_tris, 1), F = (size_t)n;
    const int slot = (c->stage_cur + 1) % (int)kStatusRing;
    for (const hg_ctx::Pending &pd : c->pw_pending_out) if (pd.stage == slot) { HG_TRY(hg_sync(c)); break; }
    for (const hg_ctx::FwdPending &pd : c->fwd_pending) if (pd.stage == slot) { HG_TRY(hg_sync(c)); break; }   // (their redo reads the staged set too)
    std::vector<FrameDesc> fresh;
    HG_TRY(fill_frames(c, fresh, geoms, offs, n));
    HG_TRY(ensure(c, c->d_set, c->set_cap, sizeof(FrameDesc) * F + sizeof(float) * 2 * c->n_pts * F));
    HG_TRY(ensure(c, c->d_trir, c->trir_cap, F * T));
    HG_TRY(ensure(c, c->d_trix, c->trix_cap, F * T));
    HG_TRY(ensure(c, c->d_segs, c->segs_cap, F * T * 3));
    HG_TRY(ensure(c, c->d_fwd, c->fwd_cap, F * T * 6));
    HG_TRY(ensure(c, c->d_inv, c->inv_cap, F * T * kInvStride));
    HG_TRY(ensure(c, c->d_status, c->status_cap, F));
    HG_TRY(ensure(c, c->d_two_round, c->two_round_cap, F));   // (never cleared: a stale word can only equal the step number by accident, which costs speed, not bits)
    if (!c->h_flag) {
        void *q = nullptr;
        hipError_t e = hipHostMalloc(&q, 64, hipHostMallocDefault);
        if (e != hipSuccess) return fail(c, HG_ERR_NOMEM, std::string("hipHostMalloc (flag word): ") + hipGetErrorString(e));
        c->h_flag = static_cast<int32_t *>(q); c->h_flag[0] = 0; c->h_flag[1] = 0;      // [0] inverse fused kernels, [1] forward tile kernels
    }
    if (F * kStatusRing > c->h_status_cap) {
        HG_TRY(hg_sync(c));
        if (c->h_status) HIP_TRY(c, hipHostFree(c->h_status));
        c->h_status = nullptr; c->h_status_cap = 0;
        void *q = nullptr;
        HIP_TRY(c, hipHostMalloc(&q, sizeof(int32_t) * F * kStatusRing, hipHostMallocDefault));
        c->h_status = static_cast<int32_t *>(q); c->h_status_cap = F * kStatusRing;
    }
    hg_ctx::Stage &st = c->stage[slot];
    const size_t fd_bytes = sizeof(FrameDesc) * F, pt_bytes = sizeof(float) * 2 * c->n_pts * F;
    // An older upload out of this slot may still be queued (no queued run refers to the slot -- checked above -- but its DMA reads
    // it): wait for THAT copy, not for the stream.  64 sets back it has long run in any loop that also launches kernels.
    if (!st.done) HIP_TRY(c, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    if (st.used) HIP_TRY(c, hipEventSynchronize(st.done));
    if (fd_bytes + pt_bytes > st.cap) {
        if (st.h) { HIP_TRY(c, hipHostFree(st.h)); st.h = nullptr; st.cap = 0; }
        void *q = nullptr;
        const size_t want = fd_bytes + pt_bytes + (fd_bytes + pt_bytes) / 4;
        hipError_t e = hipHostMalloc(&q, want, hipHostMallocDefault);
        if (e != hipSuccess) return fail(c, HG_ERR_NOMEM, std::string("hipHostMalloc (frame-set staging): ") + hipGetErrorString(e));
        st.h = static_cast<uint8_t *>(q); st.cap = want;
    }
    std::memcpy(st.h, fresh.data(), fd_bytes);
    std::memcpy(st.h + fd_bytes, dst, pt_bytes);
    st.n = n; st.n_pts = c->n_pts;
    static_assert(sizeof(FrameDesc) % 8 == 0, "the destiny points follow the frame records in the same block");
    c->d_pw_frames = reinterpret_cast<FrameDesc *>(c->d_set); c->d_dst = reinterpret_cast<float *>(c->d_set + fd_bytes);
    HG_TRY(upload_staged(c, c->d_set, st.h, fd_bytes + pt_bytes));      // (one copy: the staged block has the device layout)
    HIP_TRY(c, hipEventRecord(st.done, c->stream));
    st.used = true;
    c->pw_frames.swap(fresh);
    c->stage_cur = slot;
    double tri_rows = 0.0, shear = 0.0;
    int group_tris = 0, max_w = 0, cover = 0, tall = 0;
    double fill = 1.0;
    int64_t total_px = 0;
    for (const FrameDesc &d : c->pw_frames) { max_w = std::max(max_w, d.obj_w); if (d.obj_w > 0 && d.obj_h > 0) total_px += (int64_t)d.obj_w * d.obj_h; }
    int max_h = 0;
    for (const FrameDesc &d : c->pw_frames) if (d.obj_w > 0) max_h = std::max(max_h, d.obj_h);
    const bool quick = c->pw_quick_layout || (total_px < ((int64_t)4 << 20) && (int64_t)F * c->n_tris > 4096);
    // The host walk over every triangle of every frame only picks kernel LAYOUTS (rows per workgroup, entry format, k_pw_patch);
    // the kernels check the real counts and flag what does not fit.  A caller that uploads fresh points for the same mesh and
    // the same window shape every step (the reference's loop, test/benchmark.js:107-110) therefore keeps the previous estimate:
    // same frame count, same mesh, window extents within 1/16; re-walked every 256 sets and whenever a run had to be redone.
    hg_ctx::LayoutKey key;
    key.n = n; key.n_tris = c->n_tris; key.max_w = max_w; key.max_h = max_h; key.mesh_gen = c->mesh_gen; key.quick = quick;
    const hg_ctx::LayoutKey &ok = c->layout_key;
    const bool same_shape = ok.n == key.n && ok.n_tris == key.n_tris && ok.mesh_gen == key.mesh_gen && ok.quick == key.quick &&
                            std::abs(ok.max_w - key.max_w) * 16 <= ok.max_w && std::abs(ok.max_h - key.max_h) * 16 <= ok.max_h &&
                            c->layout_age < 256;
    if (same_shape) {
        cover = c->pw_cover; tri_rows = c->pw_tri_rows; group_tris = c->pw_group_tris; shear = c->pw_shear; tall = c->pw_tri_rows_max; fill = c->pw_fill;
        c->layout_age++;
    } else if (quick) {
        // small frames of a dense mesh (the README's 400x400 / 23 000-triangle benchmark): walking every triangle on the host
        // would cost more than the frame (and the forward paths, which only need the per-triangle solves, skip the walk); guess the row density from the triangle count (the span lists grow if it was low)
        cover = (int)(2.5 * std::sqrt((double)c->n_tris));
        group_tris = 1 << 30;                               // (no k_pw_patch without the real estimate)
        tri_rows = 64.0;
        tall = 0;                                           // (unknown: no table path without the walk)
    } else {
        cover = max_row_cover(c, dst, &tri_rows, &group_tris, &shear, &tall, &fill);
        c->pw_layout_walks++;
    }
    if (!same_shape) { c->layout_key = key; c->layout_age = 0; }
    c->pw_tri_rows = tri_rows; c->pw_group_tris = group_tris; c->pw_tri_rows_max = tall; c->pw_fill = fill;
    c->pw_cover = cover;
    c->pw_spans_per_window = max_w > 0 ? (double)cover * 256.0 / (double)max_w : 0.0;
    c->pw_row_group = cover <= 56 ? kRowGroup : 1;
    {   // few rows in total (a single 4K frame has 560 four-row groups for 256 CUs): one row per workgroup fills the chip better
        int64_t groups = 0;
        for (const FrameDesc &d : c->pw_frames) if (d.obj_w > 0 && d.obj_h > 0) groups += (d.obj_h + kRowGroup - 1) / kRowGroup;
        c->pw_small_set = groups < c->opt_min_row_groups; c->pw_groups = groups;
        if (c->pw_small_set) c->pw_row_group = 1;
    }
    // dense rows that still fit the patch kernel's LDS budget, sheared enough for 2-D gather patches to pay.  Measured
    // (k_pw_rows one row per workgroup -> k_pw_patch): C5, shear 0.39, cover 190: 0.63 -> 0.50 ms; 4K 60x60 grid, 0.15, 148:
    // 0.58 -> 0.50; 40x40, 0.16, 98: 0.47 -> 0.45; 32x32, 0.18, 92: 0.43 -> 0.42; but 24x24, 0.08, 60: 0.37 -> 0.38 and C5
    // without its shear, 0.04, 150 (5 spans per 256-pixel window): 0.35 -> 0.37.  Regardless of shear it also wins when many
    // narrow spans share a window (k_pw_rows tests every span of a window on all four pixels of every lane, k_pw_patch only
    // the spans of the lane's 64-pixel bin): lens-distortion style 64x36 grid on 4K, shear 0.02, 8.5 spans per window:
    // 0.73 -> 0.49 ms.  Layout choice only: the kernels check the real counts.
    c->pw_patch = cover > 56 && cover <= kPatchMaxRowSpans && group_tris <= kPatchMaxGroupTris && max_w <= kPatchMaxW &&
                  (int64_t)cover * 64 <= (int64_t)8 * max_w &&          // spans per 64-pixel bin ~ cover * 64 / width: overfull bins are slow
                  (shear >= 0.1 || (int64_t)cover * 256 >= (int64_t)6 * max_w) && !c->pw_patch_disabled;
    // one source per frame (decided at run time, hg_set_images_device may follow): k_pw_patch whenever the frame set fits it, see patch_preferred()
    {
        int64_t groups = 0;
        for (const FrameDesc &d : c->pw_frames) if (d.obj_w > 0 && d.obj_h > 0) groups += (d.obj_h + kRowGroup - 1) / kRowGroup;
        c->pw_patch_fits = cover <= kPatchMaxRowSpans && group_tris <= kPatchMaxGroupTris && max_w <= kPatchMaxW && max_w >= 256 &&
                           (int64_t)cover * 64 <= (int64_t)8 * max_w && groups >= c->opt_min_row_groups && !c->pw_patch_disabled;
    }
    c->pw_shear = shear;
    // beyond that budget, up to ~480 spans per row: the same kernel without matrix records in LDS (pixels read them from global)
    c->pw_patch_dense = !c->pw_patch && cover > kPatchMaxRowSpans && cover <= kPatchMaxRowSpansDense && max_w <= kPatchMaxW &&
                        (int64_t)cover * 64 <= (int64_t)8 * max_w && !c->pw_patch_disabled;
    if (c->pw_patch_dense) c->pw_patch = true;
    // k_tri_spans: one thread per triangle row and round.  Measured round 3 (step ms, 64 / 128 / 256 threads): C5 (~150 rows per
    // triangle, 8 frames) 0.475 / 0.463 / 0.489, C3 (~300 rows, 64 frames) 0.612 / 0.589 / 0.603; a single 4K frame 25.4 / 23.4 / 22.6 us
    c->pw_tri_threads = tri_rows <= 96.0 ? 64 : ((int64_t)F * c->n_tris <= 2048 && tri_rows > 128.0 ? 256 : 128);
    if (cover > 48 && c->row_cap < kRowSpanCapFast) c->row_cap = kRowSpanCapFast;   // dense rows: size the span lists up front
    if (cover > 200 && c->row_cap < kRowSpanCapDense) c->row_cap = kRowSpanCapDense;
    // (the row counters and the status ring are reused as they are when their layout -- frame count, rows per frame, list
    //  capacity, entry format -- is that of the previous set: run_setup())
    c->pw_setup_done = false;
    return HG_OK;
}

extern "C" int hg_piecewise_prepare(hg_ctx *c, const float *dst, hg_geom geom)
{
    const size_t zero = 0;
    return hg_piecewise_set_frames(c, dst, &geom, &zero, 1);
}

PwMesh mesh_of(const hg_ctx *c)
{
    PwMesh m;
    m.src_pts = c->d_src; m.tris = c->d_tris; m.n_pts = c->n_pts; m.n_tris = c->n_tris;
    m.min_src_x = c->min_src_x; m.min_src_y = c->min_src_y; m.img = c->d_img; m.W = c->W; m.H = c->H;
    m.n_imgs = c->n_imgs; m.img_stride = c->img_stride;
    return m;
}

static RowLists rows_of(const hg_ctx *c);

PwFrames frames_of(const hg_ctx *c)
{
    PwFrames f;
    f.host_flag = c->h_flag;                                 // (always armed: whether hg_sync may skip the status ring must not depend on an option that can change while runs are queued)
    f.frames = c->d_pw_frames; f.dst_pts = c->d_dst; f.trir = c->d_trir; f.trix = c->d_trix; f.segs = c->d_segs; f.fwd = c->d_fwd; f.inv = c->d_inv;
    f.status = c->status_ptr ? c->status_ptr : c->d_status; f.n_frames = (int)c->pw_frames.size();
    f.two_round = c->d_two_round; f.gen = c->pw_gen;
    int mh = 0;
    for (const FrameDesc &d : c->pw_frames) if (d.obj_w > 0) mh = std::max(mh, d.obj_h);
    f.max_obj_h = mh;
    f.row_group = c->pw_row_group;
    f.tri_threads = c->pw_tri_threads;
    f.self_spans = c->pw_self ? 1 : 0;
    f.band_ent = nullptr; f.band_cnt = nullptr; f.band_stride = 0; f.n_bands = 0; f.band_cap = 0; f.band_rows_log2 = 6;
    if (c->pw_self && c->pw_bands && c->d_rowcnt) {
        const RowLists rl = rows_of(c);
        f.band_ent = c->d_bands; f.band_cnt = rl.cnt; f.band_stride = rl.row_stride; f.n_bands = c->n_bands; f.band_cap = c->band_cap;
    }
    // k_tri_spans_grouped where the per-workgroup solves of k_tri_spans dominate the producer (measured round 3, producer us, 64 frames of
    // 4K unless noted, k_tri_spans -> grouped 16 -> 64): 512 triangles 89 -> 68 -> 68, 3200: 222 -> 151 -> 114, 4608: 342 -> 238 -> 169,
    // C5 (8 frames of 5000) 80 -> 57 -> 57; but C3 (200 triangles of 216 rows) 47.7 -> 51.2 and C4 37.1 -> 38.7: their cost is the
    // slot atomics, not the solves.  64 triangles per workgroup need >= ~1000 workgroups to fill the chip.
    { const int64_t ft = (int64_t)c->pw_frames.size() * c->n_tris;
      f.tri_group = c->opt_tri_group >= 0 ? c->opt_tri_group : ((c->n_tris >= 384 && ft >= 2048) ? (ft >= 65536 ? 64 : 16) : 0); }
    // Windows per phase, measured (C3 / C4, 64 frames): shared (cache-resident) source: 2, or 4 when a window holds
    // several spans (C4's face mesh ~4.5, C3 1.5: the longer span walk then overlaps four windows' gathers); one source per
    // frame (HBM-bound): 4 windows per phase AND fewer, deeper waves -- 12-16 KB of idle LDS per workgroup leave 5 of them on a
    // CU instead of 7 (round 3, same box: C3 0.934 -> 0.910 ms, C4 0.406 -> 0.371), where k_pw_patch does not take the frame set anyway.
    // XCD bands: fixed per XCD when every frame reads the same source and the mesh fills its window (the band's source rows then
    // stay in that XCD's L2 from frame to frame), rotating with the frame otherwise (even load; measured in hg_k_piecewise.hip)
    f.xcc_rotate = c->opt_xcc_rotate >= 0 ? (c->opt_xcc_rotate != 0) : (c->n_imgs > 1 || c->pw_fill > 1.08);
    f.xcc_log2 = c->xcc_log2; f.no_hi_bounds = c->opt_hi_bounds ? 0 : 1;
    {   // sub-bands (shared source, fixed bands, several frames): as many per XCD as it takes to bring a sub-band's share of the source to
        // ~2.2 MB -- a 4K source: 2; 1080p: none (its whole band fits the L2) -- R6.12
        const int64_t band_bytes = (int64_t)c->W * c->H * 4 >> c->xcc_log2;
        const int S = c->opt_sub_bands >= 0 ? c->opt_sub_bands : (int)std::min<int64_t>(16, (band_bytes + 2300000 - 1) / 2300000);
        f.sub_bands = (S > 1 && c->n_imgs <= 1 && !f.xcc_rotate && f.n_frames > 1) ? S : 0;
        f.sub_groups = 0;                                    // (the launchers set it for their kernel's row-group height)
    }
    f.sgpr_cap = c->n_imgs <= 1;
    f.lds_pad_kb = c->n_imgs > 1 && c->pw_row_group == kRowGroup ? (c->pw_shear >= 0.1 ? 16 : 12) : 0;
    // (self-span path: its instantiations fit 56 VGPRs / 78 SGPRs whatever the phase depth -- 8 workgroups per CU where the list-reading
    //  4-window form admits 7.  Same box, alternating order, 2 -> 4 windows per phase: C3 0.5697 -> 0.5685 ms, C4 0.2220 -> 0.2228, 512-triangle
    //  grid 0.6323 -> 0.6267: a wash to a slight gain, so one depth for every self-span set; EXPERIMENTS.md R4.10)
    // Span flags "both end pixels inside the source window" (k_pw_rows: windows made of flagged spans skip the per-pixel bounds test).  Same box,
    // alternating order, off -> on: C3 (1.5 spans per 256-pixel window) 0.562 -> 0.525 ms; C4 (4.5: one flag per span piece in the prologue, few
    // windows wholly inside flagged spans) 0.217 -> 0.220.  Hence by the host's spans-per-window estimate; option "safe_spans" forces either.
    f.safe_spans = c->opt_safe_spans >= 0 ? c->opt_safe_spans : (c->pw_spans_per_window < 3.0 ? 1 : 0);
    f.safe_spans_patch = c->opt_safe_spans >= 0 ? c->opt_safe_spans : 1;
    f.phase = c->opt_phase > 0 ? c->opt_phase : (c->n_imgs > 1 || c->pw_self ? 4 : (c->pw_spans_per_window >= 3.0 ? 4 : 2));
    return f;
}

static RowLists rows_of(const hg_ctx *c)
{
    RowLists r;
    r.ent = c->d_rowent; r.cap = c->row_cap; r.compact = c->pw_compact ? 1 : 0;
    int mh = 0;
    for (const FrameDesc &d : c->pw_frames) if (d.obj_w > 0) mh = std::max(mh, d.obj_h);
    r.row_stride = std::max(mh, 1);
    const size_t set = c->pw_frames.size() * (size_t)r.row_stride;           // two counter sets, then the status ring
    r.cnt = c->d_rowcnt ? c->d_rowcnt + (size_t)c->rows_parity * set : nullptr;
    r.cnt_clear = c->d_rowcnt ? c->d_rowcnt + (size_t)(1 - c->rows_parity) * set : nullptr;
    return r;
}

// Would the next fused warp of this frame set go through k_pw_patch (the parity tap, which passes a map, never does)?
static bool patch_preferred(const hg_ctx *c, bool *global_records)
{
    const int force = c->opt_patch;                          // option "patch": -1 by estimate
    int mw = 0;
    for (const FrameDesc &d : c->pw_frames) mw = std::max(mw, d.obj_w);
    if (global_records) *global_records = force == 2 ? true : (force == 1 ? false : c->pw_patch_dense);
    // by estimate (dense, sheared rows), and -- measured round 3 -- whenever every frame streams its own source from HBM and the
    // set fits the kernel: its 16 x 4 gather patches and 8-byte lists beat k_pw_rows there even on sparse meshes (same box, one
    // source per frame: C4 0.371 -> 0.330 ms, C3 step 0.973 -> 0.947)
    // ... and -- round 4 -- whenever the rows are too dense for k_pw_rows<SELF> (more than 56 spans) and the set fits: what k_pw_rows would gain on
    // flat dense meshes (round 3, kernel only: 24x24 grid 0.38 -> 0.37 ms) is less than the span producer it needs and k_pw_patch<SELF> does
    // not (same box, step ms, alternating order: 24x24 grid on 4K 0.760 -> 0.685, C5's mesh at 1/10, 1/8, 1/4 of its shear 0.383 -> 0.351, 0.390 -> 0.355, 0.405 -> 0.363)
    return c->pw_fast && mw <= kPatchMaxW && !c->pw_patch_disabled &&
           (force >= 0 ? force >= 1 : (c->pw_patch || (c->pw_patch_fits && (c->n_imgs > 1 || c->pw_cover > 56))));
}

// per-frame solves; status words are reset first.  Fast path: k_tri_spans (solves + per-row span lists);
// general path (more than 32767 triangles, huge sources, negative source minimum): k_tri_setup.
static int run_setup(hg_ctx *c, bool for_tap = false)
{
    const size_t F = c->pw_frames.size();
    int mw = 0;
    for (const FrameDesc &d : c->pw_frames) mw = std::max(mw, d.obj_w);
    c->pw_fast = pw_fast_ok(mesh_of(c), mw);
    if (c->pw_fast) {
        // entry format of the span lists (hg_kernels.h): 8 bytes for dense rows and whenever k_pw_patch will read them
        bool global_records = false;
        const bool want_patch = patch_preferred(c, &global_records) && !for_tap;     // (the parity tap passes a map: never k_pw_patch)
        const bool compact = c->opt_compact >= 0 ? c->opt_compact != 0 : (want_patch || c->pw_cover > 56);
        if (compact != c->pw_compact) { c->pw_compact = compact; c->rows_clean = false; }
        // Self-span path (round 4): k_tri_setup in front (solves, edge equations, row reach), and the row workgroups of the warp kernel
        // evaluate the spans of their own rows in their prologue.  No span producer kernel, no lists, no slot atomics.
        //   * k_pw_rows<SELF>: sparse meshes -- where the row lists would carry 32-byte entries -- whose triangles a workgroup can afford
        //     to scan (every row group tests all of them, 16 bytes each);
        //   * k_pw_patch<SELF>: whatever that kernel takes (dense / sheared meshes, one source per frame); beyond 256 triangles the
        //     workgroups scan the candidate band k_tri_setup filed for their rows instead of the whole mesh.
        // Layout choice only: what does not fit (more candidate triangles than the LDS list, more spans per row than a block holds, an
        // overfull band) flags its frame -> map path, and the context returns to row lists for this mesh.  The prologue's int32 arithmetic
        // wants windows below 2^24 rows near the origin.
        bool small_geom = true;
        int max_h = 0;
        for (const FrameDesc &d : c->pw_frames) {
            if (d.obj_w > 0 && d.obj_h > 0 && (d.obj_h > (1 << 24) || d.obj_w < 16 || std::abs((int64_t)d.y_off) > (1 << 26))) small_geom = false;
            if (d.obj_w > 0) max_h = std::max(max_h, d.obj_h);
        }
        // Policy (measured round 4, same box, row lists -> own spans): C3 x 64 frames step 0.624 -> 0.588 ms (warp kernel + 7 us, the 47 us
        // producer gone), C4 0.258 -> 0.236-0.242, 8 frames of C3 0.104 -> 0.093; a single 4K frame 22.8 -> 23.1 us per queued step (one
        // row per workgroup: every one of 2239 workgroups scans all triangles, and k_tri_setup's 4 us are one workgroup's dependent
        // chain): small frame sets keep the row lists unless the option forces it.
        // The small / large boundary (option "min_row_groups", 1152 four-row groups), queued steps, alternating order: C3 2 frames (1120 groups)
        // lists 0.0375 vs self 0.0389, 3 frames (1680) 0.0495 vs 0.0474; C4 2 frames (624) 0.0275 vs 0.0283, 4 frames (1248) 0.0374 vs 0.0332.
        // (k_pw_patch takes the self-span form from half the threshold: a single 8K frame of C5, 1120 groups, 0.0914 -> 0.084 ms per queued step)
        const bool self_ok = !c->pw_self_disabled && small_geom &&
                             (c->opt_self >= 0 ? c->opt_self == 1 : (!c->pw_small_set || (want_patch && c->pw_groups * 2 >= c->opt_min_row_groups)));
        const bool self_patch = self_ok && want_patch && !global_records && (c->n_tris <= 256 || c->pw_tri_rows_max > 0);
        // (beyond 1024 triangles a row group scans its candidate band instead of the whole mesh, like k_pw_patch<SELF>)
        bool self_rows = self_ok && !want_patch && !compact && c->row_cap <= kRowSpanCapFast && (c->n_tris <= 1024 || (c->n_tris <= 8192 && c->pw_tri_rows_max > 0)) &&
                         (c->pw_row_group == 1 || c->pw_cover <= 56);
        c->pw_self_patch = self_patch;
        // k_pw_tile instead of k_pw_patch<SELF> where every frame streams its own source (option "tile" forces either).  Same box, one source
        // per frame, alternating order, patch -> tile (EXPERIMENTS.md R4.7): C5 0.5915 -> 0.5093 ms, its mesh at 3/4, 1/2, 1/4 of the shear
        // 0.5345 -> 0.4737, 0.486 -> 0.4528, 0.437 -> 0.429; C3 0.8155 -> 0.816, C4 0.332 -> 0.330, 40x40 / 64x36 grids 0.870 -> 0.868, 0.975 ->
        // 0.950.  With a shared source it is a toss-up (C5 0.4009 -> 0.3975, 64x36 grid 0.848 -> 0.830, 40x40 grid 0.734 -> 0.747): k_pw_patch stays.
        // Round 6 (R6.4): with the block slopes evaluated once per tile the tile kernel also wins on a SHARED source where the mesh is steeply
        // sheared or packs more than two spans into a 64-pixel block, k_pw_patch -> k_pw_tile: C5 (shear 0.39) 0.395 -> 0.378 ms, 64x36 grid
        // (128 spans per 3840-pixel row) 0.834 -> 0.783; not elsewhere: 40x40 grid 0.721 -> 0.703 but 24x24 0.642 -> 0.638, C5's mesh at 3/8
        // and 1/8 of its shear 0.353 -> 0.352 and 0.331 -> 0.338, 20x60 tall cells 0.631 -> 0.636.
        const bool tile_shared = c->pw_shear >= 0.3 || (int64_t)c->pw_cover * 64 >= (int64_t)2 * mw;
        // ... and, on a shared source, only where the host's estimate says a tile row holds its spans (a tile is at most 2048 columns of the row: the
        // estimate -- an upper bound, it counts every piece of the densest row -- split evenly; the kernel checks the real counts and flags what does
        // not fit, which is how one source per frame finds out: there the redo is paid once and the mesh goes back to k_pw_patch)
        const int col_tiles = std::max(1, (mw + 2047) / 2048);
        const bool tile_fits = (c->pw_cover + col_tiles - 1) / col_tiles <= kTileRowSpanCap;
        c->pw_tile = self_patch && !c->pw_tile_disabled && mw >= 512 && (c->opt_tile >= 0 ? c->opt_tile == 1 : (c->n_imgs > 1 || (tile_shared && tile_fits)));
        c->pw_bands = (self_patch && c->n_tris > 256) || (self_rows && c->n_tris > 1024);
        if (c->pw_bands && (std::max(max_h, 1) + 63) / 64 > 2048) {      // (kBandMax; frames taller than 131 072 rows)
            c->pw_bands = false; c->pw_self_patch = false;
            if (c->n_tris > 1024) self_rows = false;
        }
        if (c->pw_bands) {
            // bands of 64 output rows; capacity from the tallest triangle of the frame set (host estimate; an overfull band flags its frame)
            c->n_bands = (std::max(max_h, 1) + 63) / 64;
            const double per_band = (double)c->n_tris * ((double)c->pw_tri_rows_max + 64.0 + 8.0) / (double)std::max(max_h, 64);
            int cap = (int)std::min<double>((double)c->n_tris, std::max(256.0, 2.0 * per_band));
            cap = (cap + 63) & ~63;
            if (cap > c->band_cap) c->band_cap = cap;
            const size_t need = F * (size_t)c->n_bands * (size_t)c->band_cap * 2;      // (two int4 per entry)
            if (need > c->bands_cap) HG_TRY(hg_sync(c));
            HG_TRY(ensure(c, c->d_bands, c->bands_cap, need));
        }
        const bool self = c->pw_self_patch || self_rows;
        // (its warp kernel neither reads nor zeroes the row counters: a change of path starts from freshly zeroed counter sets)
        if (self != c->pw_self) { c->pw_self = self; c->rows_clean = false; }
        RowLists rl = rows_of(c);
        // Two sets of row counters (ping-pong) + kStatusRing sets of per-frame status words share one allocation.  It is zeroed by a
        // memset only when the layout changes (or after a setup whose warp never ran): the warp kernel of a step zeroes the OTHER
        // counter set -- the one the previous step consumed, the one the next step counts into -- and the next status set.
        const int32_t *before = c->d_rowcnt;
        const size_t ent_bytes = c->pw_self ? 0 : F * (size_t)rl.row_stride * rl.cap * (c->pw_compact ? sizeof(RowEnt8) : sizeof(RowEnt));
        if (2 * F * rl.row_stride + kStatusRing * F > c->rowcnt_cap || ent_bytes > c->rowent_cap) HG_TRY(hg_sync(c));   // (queued runs flag into the old ring)
        HG_TRY(ensure(c, c->d_rowcnt, c->rowcnt_cap, 2 * F * rl.row_stride + kStatusRing * F));
        HG_TRY(ensure(c, c->d_rowent, c->rowent_cap, ent_bytes));
        rl = rows_of(c);
        if (before != c->d_rowcnt || c->rows_F != F || c->rows_stride != rl.row_stride || c->rows_cap != rl.cap) c->rows_clean = false;
        c->rows_F = F; c->rows_stride = rl.row_stride; c->rows_cap = rl.cap;
        if (c->rows_clean) { c->status_slot = (c->status_slot + 1) % (int)kStatusRing; c->rows_parity ^= 1; }
        else {
            HG_TRY(hg_sync(c));                              // (queued runs still own status sets)
            c->status_slot = 0; c->rows_parity = 0;
            HIP_TRY(c, hipMemsetAsync(c->d_rowcnt, 0, sizeof(int32_t) * (2 * F * rl.row_stride + kStatusRing * F), c->stream));
        }
        rl = rows_of(c);                                     // (parity settled)
        c->status_base = c->d_rowcnt + 2 * F * rl.row_stride;
        c->status_ptr = c->status_base + (size_t)c->status_slot * F;
        c->status_next = c->status_base + (size_t)((c->status_slot + 1) % (int)kStatusRing) * F;
        c->rows_clean = false;                               // dirty until the warp kernel has run: it consumes the counters and clears the next status set
        if (c->pw_self) c->pw_gen = c->pw_gen >= 0x7ffffff0 ? 1 : c->pw_gen + 1;   // (PwFrames::gen: k_tri_setup and the warp kernel behind it see the same number)
        if (c->pw_self) launch_tri_setup(mesh_of(c), frames_of(c), c->stream);     // solves, edge equations, row reach (+ candidate bands, counted in the row counters' place)
        else            launch_tri_spans(mesh_of(c), frames_of(c), rl, c->stream);
    } else {
        HG_TRY(hg_sync(c));                                  // (queued fast-path runs are settled against their own status ring first)
        // the general path has no row lists, no self-span prologue and no candidate bands: frames_of() must not hand k_tri_setup the
        // band buffers an earlier fast-path set was laid out for (sized for ITS frame count and height), and the next fast-path set
        // starts from freshly zeroed counters
        c->pw_self = false; c->pw_bands = false; c->pw_self_patch = false; c->pw_tile = false; c->rows_clean = false;
        c->status_ptr = c->d_status;
        HIP_TRY(c, hipMemsetAsync(c->d_status, 0, sizeof(int32_t) * F, c->stream));
        launch_tri_setup(mesh_of(c), frames_of(c), c->stream);
    }
    HIP_TRY(c, hipGetLastError());
    c->pw_setup_done = true;
    return HG_OK;
}

static void run_warp(hg_ctx *c, uint8_t *d_out, int16_t *map_out)
{
    bool global_records = false;
    const bool patch = c->pw_self ? (c->pw_self_patch && !map_out) : (patch_preferred(c, &global_records) && !map_out && c->pw_compact);
    c->pw_used_patch = patch;
    c->pw_last_kernel = patch ? 3 : (c->pw_fast ? (c->pw_row_group == kRowGroup ? 1 : 2) : 4);
    if (patch && c->pw_self && c->pw_tile) {
        int mw = 0;
        for (const FrameDesc &d : c->pw_frames) mw = std::max(mw, d.obj_w);
        c->pw_last_kernel = 5;
        c->pw_used_patch = false;                            // (a flagged tile run disables k_pw_tile for the mesh, not k_pw_patch)
        c->pw_last_variant = launch_pw_tile(mesh_of(c), frames_of(c), rows_of(c), d_out, mw, c->status_next, c->stream); c->rows_clean = true;
    }
    else if (patch)      { c->pw_last_variant = launch_pw_patch(mesh_of(c), frames_of(c), rows_of(c), d_out, c->status_next, global_records, c->stream); c->rows_clean = true; }
    else if (c->pw_fast) {
        c->pw_last_variant = launch_pw_rows(mesh_of(c), frames_of(c), rows_of(c), d_out, map_out, c->status_next, c->stream); c->rows_clean = true;
    }
    else          { launch_pw_fused(mesh_of(c), frames_of(c), d_out, map_out, c->stream); c->pw_last_variant = 600000; }
}

static int check_pw_state(hg_ctx *c)
{
    if (!c->d_img) return fail(c, HG_ERR_STATE, "no source image: call hg_set_image first");
    if (!c->have_mesh) return fail(c, HG_ERR_STATE, "no mesh: call hg_piecewise_set_mesh first");
    if (c->pw_frames.empty()) return fail(c, HG_ERR_STATE, "no frame prepared: call hg_piecewise_prepare / hg_piecewise_set_frames first");
    return HG_OK;
}

// one frame through the materialised map (exact for any input)
static int run_frame_via_map(hg_ctx *c, int f, uint8_t *d_out)
{
    const FrameDesc &fd = c->pw_frames[f];
    const size_t n = (fd.obj_w > 0 && fd.obj_h > 0) ? (size_t)fd.obj_w * fd.obj_h : 0;
    if (n == 0) return HG_OK;
    HG_TRY(ensure(c, c->d_map32, c->map32_cap, n));
    PwMesh mesh = mesh_of(c);
    mesh.img = frame_img(mesh, f); mesh.n_imgs = 1;          // this frame's own source
    launch_map_build(mesh, frames_of(c), f, fd, c->d_map32, c->stream);
    launch_pw_from_map(mesh, frames_of(c), f, fd, c->d_map32, d_out, c->stream);
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

// Deferred redo: frame f of the staged set `stage` (the set a queued run warped; newer sets may have been uploaded since)
// through the materialised map, into `d_out` at the frame's own offset.  Self-contained: the frame's window and points go from
// the staging buffer to a one-frame scratch, k_tri_setup solves it there, then rasteriser + pixel loop.  Mesh and source
// image are those of the context (changing either settles queued runs first).
static int redo_frame_staged(hg_ctx *c, int stage, int f, uint8_t *d_out)
{
    const hg_ctx::Stage &st = c->stage[stage];
    if (stage < 0 || !st.h || f >= st.n || st.n_pts != c->n_pts) return fail(c, HG_ERR_STATE, "deferred redo: the staged frame set is gone");
    const FrameDesc fd = reinterpret_cast<const FrameDesc *>(st.h)[f];
    const size_t n = (fd.obj_w > 0 && fd.obj_h > 0) ? (size_t)fd.obj_w * fd.obj_h : 0;
    if (n == 0) return HG_OK;
    const size_t T = (size_t)std::max(c->n_tris, 1);
    HG_TRY(ensure(c, c->d_redo_frame, c->redo_frame_cap, (size_t)1));
    HG_TRY(ensure(c, c->d_redo_dst, c->redo_dst_cap, (size_t)c->n_pts * 2));
    HG_TRY(ensure(c, c->d_redo_trir, c->redo_trir_cap, T));
    HG_TRY(ensure(c, c->d_redo_trix, c->redo_trix_cap, T));
    HG_TRY(ensure(c, c->d_redo_segs, c->redo_segs_cap, T * 3));
    HG_TRY(ensure(c, c->d_redo_fwd, c->redo_fwd_cap, T * 6));
    HG_TRY(ensure(c, c->d_redo_inv, c->redo_inv_cap, T * kInvStride));
    HG_TRY(ensure(c, c->d_redo_status, c->redo_status_cap, (size_t)1));
    HG_TRY(ensure(c, c->d_map32, c->map32_cap, n));
    const float *pts = reinterpret_cast<const float *>(st.h + sizeof(FrameDesc) * (size_t)st.n) + (size_t)f * c->n_pts * 2;
    HIP_TRY(c, hipMemcpyAsync(c->d_redo_frame, st.h + sizeof(FrameDesc) * (size_t)f, sizeof(FrameDesc), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_redo_dst, pts, sizeof(float) * 2 * c->n_pts, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_redo_status, 0, sizeof(int32_t), c->stream));
    PwMesh mesh = mesh_of(c);
    mesh.img = frame_img(mesh, f); mesh.n_imgs = 1;          // this frame's own source
    PwFrames fr = frames_of(c);
    fr.two_round = nullptr;                                  // (a set-up of its own, outside the frame set's step numbering)
    fr.frames = c->d_redo_frame; fr.dst_pts = c->d_redo_dst; fr.trir = c->d_redo_trir; fr.trix = c->d_redo_trix; fr.band_ent = nullptr; fr.host_flag = nullptr; fr.segs = c->d_redo_segs; fr.fwd = c->d_redo_fwd;
    fr.inv = c->d_redo_inv; fr.status = c->d_redo_status; fr.n_frames = 1; fr.max_obj_h = fd.obj_h;
    launch_tri_setup(mesh, fr, c->stream);
    launch_map_build(mesh, fr, 0, fd, c->d_map32, c->stream);
    launch_pw_from_map(mesh, fr, 0, fd, c->d_map32, d_out, c->stream);
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

// The forward counterpart: frame f of the staged set through k_fwd_scatter_pw + k_fwd_gather with its own matrices (solved in the
// one-frame scratch), over the context's forward map.
int redo_forward_frame_staged(hg_ctx *c, int stage, int f, int max_src_x, int max_src_y, uint8_t *d_out)
{
    if (stage < 0 || !c->fmap_valid) return fail(c, HG_ERR_STATE, "deferred forward redo: the staged frame set is gone");
    const hg_ctx::Stage &st = c->stage[stage];
    if (!st.h || f >= st.n || st.n_pts != c->n_pts) return fail(c, HG_ERR_STATE, "deferred forward redo: the staged frame set is gone");
    const FrameDesc fd = reinterpret_cast<const FrameDesc *>(st.h)[f];
    const size_t n = (fd.obj_w > 0 && fd.obj_h > 0) ? (size_t)fd.obj_w * fd.obj_h : 0;
    if (n == 0) return HG_OK;
    const size_t T = (size_t)std::max(c->n_tris, 1);
    HG_TRY(ensure(c, c->d_redo_frame, c->redo_frame_cap, (size_t)1));
    HG_TRY(ensure(c, c->d_redo_dst, c->redo_dst_cap, (size_t)c->n_pts * 2));
    HG_TRY(ensure(c, c->d_redo_trir, c->redo_trir_cap, T));
    HG_TRY(ensure(c, c->d_redo_trix, c->redo_trix_cap, T));
    HG_TRY(ensure(c, c->d_redo_segs, c->redo_segs_cap, T * 3));
    HG_TRY(ensure(c, c->d_redo_fwd, c->redo_fwd_cap, T * 6));
    HG_TRY(ensure(c, c->d_redo_inv, c->redo_inv_cap, T * kInvStride));
    HG_TRY(ensure(c, c->d_redo_status, c->redo_status_cap, (size_t)1));
    HG_TRY(ensure(c, c->d_win32, c->win32_cap, n));
    const float *pts = reinterpret_cast<const float *>(st.h + sizeof(FrameDesc) * (size_t)st.n) + (size_t)f * c->n_pts * 2;
    HIP_TRY(c, hipMemcpyAsync(c->d_redo_frame, st.h + sizeof(FrameDesc) * (size_t)f, sizeof(FrameDesc), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_redo_dst, pts, sizeof(float) * 2 * c->n_pts, hipMemcpyHostToDevice, c->stream));
    PwFrames fr = frames_of(c);
    fr.two_round = nullptr;                                  // (a set-up of its own, outside the frame set's step numbering)
    fr.frames = c->d_redo_frame; fr.dst_pts = c->d_redo_dst; fr.trir = c->d_redo_trir; fr.trix = c->d_redo_trix; fr.band_ent = nullptr; fr.host_flag = nullptr; fr.segs = c->d_redo_segs; fr.fwd = c->d_redo_fwd;
    fr.inv = c->d_redo_inv; fr.status = c->d_redo_status; fr.n_frames = 1; fr.max_obj_h = fd.obj_h;
    launch_tri_setup(mesh_of(c), fr, c->stream);
    launch_fwd_pw(c->d_fmap, c->d_redo_fwd, frame_img(mesh_of(c), f), c->W, c->H, c->min_src_x, c->min_src_y, max_src_x - c->min_src_x, max_src_y - c->min_src_y,
                  fd, c->d_win32, d_out, c->stream);
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

extern "C" int hg_warp_inverse_piecewise_frames_device(hg_ctx *c, void *d_out)
{
    HG_TRY(bind(c));
    if (!d_out) return fail(c, HG_ERR_INVALID, "d_out is NULL");
    HG_TRY(check_pw_state(c));
    if (c->pw_pending_out.size() >= kStatusRing - 1 || !c->fwd_pending.empty()) HG_TRY(hg_sync(c));
    size_t extent = 0;
    uint64_t layout = 0;
    output_layout(c->pw_frames, &extent, &layout);           // (hg_sync orders the deferred redos of overlapping runs: replay_queued)
    // The reference recomputes the per-triangle matrices on every setDestinyPoints and the map + inverses on every
    // warp(): both are part of the per-frame step, so both run here every time.
    HG_TRY(run_setup(c));
    HG_TRY(time_begin(c));
    run_warp(c, static_cast<uint8_t *>(d_out), nullptr);
    HG_TRY(time_end(c));
    HIP_TRY(c, hipGetLastError());
    const uint8_t path = (uint8_t)((c->pw_used_patch ? 1 : 0) | (c->pw_self ? 2 : 0) | (c->pw_self && c->pw_tile && c->pw_last_kernel == 5 ? 4 : 0));
    if (c->pw_fast) c->pw_pending_out.push_back({static_cast<uint8_t *>(d_out), c->status_slot, c->stage_cur, extent, layout, path});
    else {                                                   // general path: one status set, checked right away
        HIP_TRY(c, hipMemcpyAsync(c->h_status, c->status_ptr, sizeof(int32_t) * c->pw_frames.size(), hipMemcpyDeviceToHost, c->stream));
        c->status_base = nullptr;
        c->pw_pending_out.push_back({static_cast<uint8_t *>(d_out), 0, c->stage_cur, extent, layout, 0});
        HG_TRY(hg_sync(c));
    }
    return HG_OK;
}

// ---- settling queued runs in call order (hg_sync)
// The bytes frame f of a queued run wrote: from the run's staged frame set (the context's current one if the run had none).
struct FrameRange { const uint8_t *lo, *hi; };
static bool queued_frame_range(const hg_ctx *c, int stage, int f, const uint8_t *out, FrameRange *r)
{
    const FrameDesc *fd = nullptr;
    if (stage >= 0 && c->stage[stage].h) { if (f < c->stage[stage].n) fd = reinterpret_cast<const FrameDesc *>(c->stage[stage].h) + f; }
    else if (f < (int)c->pw_frames.size()) fd = &c->pw_frames[f];
    if (!fd || fd->obj_w <= 0 || fd->obj_h <= 0) return false;
    r->lo = out + fd->out_off; r->hi = r->lo + (size_t)fd->obj_w * fd->obj_h * 4;
    return true;
}

// Sequential semantics for deferred redos: runs are visited in call order; a frame is redone if the device flagged it, or if it
// overlaps bytes an EARLIER run's redo has just rewritten (that redo came after this frame's fast-path write, so the frame is put
// back on top of it) -- unless a LATER queued run wrote exactly the same byte range (a caller reusing one buffer with one layout:
// the later frame is the newer one, the older redo is skipped).  n_frames(i), flagged(i, f), redo(i, f) describe the list.
template <class P, class NF, class Flagged, class Redo>
static int replay_queued(hg_ctx *c, const std::vector<P> &pending, NF n_frames, Flagged flagged, Redo redo)
{
    std::vector<FrameRange> dirty;
    for (size_t i = 0; i < pending.size(); i++) {
        const P &p = pending[i];
        for (int f = 0; f < n_frames(i); f++) {
            FrameRange r;
            if (!queued_frame_range(c, p.stage, f, p.out, &r)) continue;
            bool again = flagged(i, f);
            for (size_t d = 0; d < dirty.size() && !again; d++) again = dirty[d].lo < r.hi && r.lo < dirty[d].hi;
            if (!again) continue;
            bool superseded = false;
            for (size_t j = i + 1; j < pending.size() && !superseded; j++) {
                const P &q = pending[j];
                if (q.out == p.out && q.extent == p.extent && q.layout == p.layout) { superseded = true; break; }     // same buffer, same layout
                FrameRange w;
                for (int g = 0; g < n_frames(j) && !superseded; g++)
                    superseded = queued_frame_range(c, q.stage, g, q.out, &w) && w.lo == r.lo && w.hi == r.hi;
            }
            if (superseded) continue;
            HG_TRY(redo(i, f));
            dirty.push_back(r);
        }
    }
    return HG_OK;
}

extern "C" int hg_sync(hg_ctx *c)
{
    HG_TRY(bind(c));
    // Fast-path runs set the page-locked word h_flag whenever a kernel flags a frame (flag_frame, hg_dev.h): if it is still 0 after the
    // synchronisation nobody flagged anything and the status ring is not read at all; otherwise it comes down with a blocking copy (rare).
    // (A blocking hipMemcpy after EVERY synchronisation made a synced single 4K frame 47.6 us on the host; a download kernel queued
    //  behind the runs 36.2; this, 34.0: EXPERIMENTS.md R4.13.)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!c->pw_pending_out.empty()) {
        // frames a fused run flagged (irregular, or a row list overflowed) are redone through the materialised map, into the
        // output of the call that flagged them, in call order
        std::vector<hg_ctx::Pending> pending;
        pending.swap(c->pw_pending_out);
        bool redo = false;
        // (all queued runs share one layout of the status ring: a set with another frame count settles them before it runs)
        const int st0 = pending.front().stage;
        const size_t F = (st0 >= 0 && c->stage[st0].h) ? (size_t)c->stage[st0].n : c->pw_frames.size();
        const bool none_flagged = c->status_base && c->h_flag && *c->h_flag == 0;
        if (c->status_base && !none_flagged) {
            HIP_TRY(c, hipMemcpy(c->h_status, c->status_base, sizeof(int32_t) * F * kStatusRing, hipMemcpyDeviceToHost));
            if (c->h_flag) *c->h_flag = 0;                   // (the GPU is idle here)
        }
        uint8_t over_paths = 0;                               // paths (Pending::path bits) of the runs that exceeded a kernel LIMIT (not merely irregular input)
        if (!none_flagged) for (size_t i = 0; i < pending.size(); i++)
            for (size_t f = 0; f < F; f++) {
                const int32_t w = c->h_status[(size_t)pending[i].slot * F + f];
                if (w == FRAME_OK) continue;
                redo = true; c->pw_redone++; c->pw_last_flag = w;
                if (w & FRAME_LDS_OVERFLOW) over_paths |= (uint8_t)(pending[i].path | 8);
            }
        if (redo)
            HG_TRY(replay_queued(c, pending, [&](size_t) { return (int)F; },
                                 [&](size_t i, int f) { return c->h_status[(size_t)pending[i].slot * F + f] != FRAME_OK; },
                                 [&](size_t i, int f) { return redo_frame_staged(c, pending[i].stage, f, pending[i].out); }));
        if (redo) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            // A run that exceeded a LIMIT of its kernel teaches the layout policy, by the path THAT run took; a frame that was merely
            // irregular (NaN / absurd vertices: FRAME_IRREGULAR alone) goes through the map path and teaches nothing.
            if (over_paths & 8) {
                if (over_paths & 1) c->pw_patch_disabled = true; // k_pw_patch (row lists or self-spans): its limits are tighter than k_pw_rows'
                if (c->row_cap < kRowSpanCapDense)                // denser mesh than assumed: larger lists next time
                    c->row_cap = c->row_cap < kRowSpanCapFast ? kRowSpanCapFast : kRowSpanCapDense;
                if (over_paths & 4) c->pw_tile_disabled = true;  // a tile beyond its limits: k_pw_patch<SELF> for this mesh
                else if (over_paths & 2) c->pw_self_disabled = true;   // the self-span path: more candidates / spans than its LDS blocks hold -> row lists for this mesh
            }
            c->layout_age = 1 << 30;                             // ... and a fresh layout estimate for the next frame set
        }
    }
    if (!c->fwd_pending.empty()) {
        // tile-binned forward piecewise frames the device flagged (a triangle it could not bound, an overfull tile list): redone
        // through scatter + gather into the output of the call that flagged them, from that call's staged frame set (newer sets
        // may have been uploaded since).  The forward map is the context's: a new mesh settles queued runs first.
        std::vector<hg_ctx::FwdPending> pending;
        pending.swap(c->fwd_pending);
        std::vector<int32_t> st(c->fwd_status_cap);
        // (as above: the status words are read only when a forward tile kernel set its host-visible flag word)
        const bool none_flagged = c->h_flag && c->h_flag[1] == 0;
        if (!none_flagged) {
            HIP_TRY(c, hipMemcpy(st.data(), c->d_fwd_status, sizeof(int32_t) * st.size(), hipMemcpyDeviceToHost));
            if (c->h_flag) c->h_flag[1] = 0;
        }
        bool overflow = false, unbounded = false, any = false;
        if (!none_flagged) for (const hg_ctx::FwdPending &fp : pending) {
            const int32_t *sf = st.data() + (size_t)fp.slot * c->fwd_status_stride;
            for (int f = 0; f < fp.n; f++) {
                if (sf[f] == 0) continue;
                any = true;
                if (sf[f] & FWD_OVERFLOW) overflow = true;
                if (sf[f] & FWD_FALLBACK) unbounded = true;
            }
        }
        if (any)
            HG_TRY(replay_queued(c, pending, [&](size_t i) { return pending[i].n; },
                                 [&](size_t i, int f) { return st[(size_t)pending[i].slot * c->fwd_status_stride + f] != 0; },
                                 [&](size_t i, int f) { c->pw_redone++;
                                                        return redo_forward_frame_staged(c, pending[i].stage, f, pending[i].max_src_x, pending[i].max_src_y, pending[i].out); }));
        if (any) {
            HIP_TRY(c, hipMemsetAsync(c->d_fwd_status, 0, sizeof(int32_t) * c->fwd_status_cap, c->stream));   // (zero between calls)
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        if (overflow) {
            if (c->fwd_pw_cap < kFwdPwCapMax) c->fwd_pw_cap = std::min(kFwdPwCapMax, c->fwd_pw_cap * 2);
            else c->fwd_pw_tiles_disabled = true;
        }
        if (unbounded) c->fwd_pw_tiles_disabled = true;      // (a degenerate triangle in this mesh: do not pay for both paths again)
    }
    const int d = c->deferred;
    c->deferred = HG_OK;
    return d;
}

extern "C" int hg_warp_inverse_piecewise_batch_device(hg_ctx *c, const float *dst, const hg_geom *geoms, const size_t *offs, int n, void *d_out)
{
    HG_TRY(hg_piecewise_set_frames(c, dst, geoms, offs, n));
    return hg_warp_inverse_piecewise_frames_device(c, d_out);
}

extern "C" int hg_warp_inverse_piecewise_device(hg_ctx *c, void *d_out) { return hg_warp_inverse_piecewise_frames_device(c, d_out); }

static int single_frame_bytes(hg_ctx *c, size_t *bytes)
{
    HG_TRY(check_pw_state(c));
    if (c->pw_frames.size() != 1) return fail(c, HG_ERR_STATE, "this call needs exactly one prepared frame (hg_piecewise_prepare)");
    const FrameDesc &fd = c->pw_frames[0];
    *bytes = (fd.obj_w > 0 && fd.obj_h > 0) ? (size_t)fd.obj_w * fd.obj_h * 4 : 0;
    return HG_OK;
}

extern "C" int hg_warp_inverse_piecewise(hg_ctx *c, uint8_t *out_host)
{
    HG_TRY(bind(c));
    if (!out_host) return fail(c, HG_ERR_INVALID, "out is NULL");
    size_t bytes = 0;
    HG_TRY(single_frame_bytes(c, &bytes));
    if (bytes == 0) return HG_OK;
    const uint64_t keep = c->pw_frames[0].out_off;
    if (keep != 0) return fail(c, HG_ERR_STATE, "prepared frame has a non-zero output offset");
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, bytes));
    HG_TRY(hg_warp_inverse_piecewise_frames_device(c, c->d_out_tmp));
    HG_TRY(hg_sync(c));
    HIP_TRY(c, hipMemcpy(out_host, c->d_out_tmp, bytes, hipMemcpyDeviceToHost));
    return HG_OK;
}

extern "C" int hg_warp_inverse_piecewise_via_map(hg_ctx *c, uint8_t *out_host)
{
    HG_TRY(bind(c));
    if (!out_host) return fail(c, HG_ERR_INVALID, "out is NULL");
    size_t bytes = 0;
    HG_TRY(single_frame_bytes(c, &bytes));
    if (bytes == 0) return HG_OK;
    HG_TRY(hg_sync(c));
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, bytes));
    HG_TRY(run_setup(c));
    HG_TRY(run_frame_via_map(c, 0, c->d_out_tmp));
    HIP_TRY(c, hipMemcpyAsync(out_host, c->d_out_tmp, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HG_OK;
}

extern "C" int hg_get_tri_map(hg_ctx *c, int16_t *out, size_t len)
{
    HG_TRY(bind(c));
    size_t bytes = 0;
    HG_TRY(single_frame_bytes(c, &bytes));
    const size_t n = bytes / 4;
    if (!out || len != n) return fail(c, HG_ERR_INVALID, "hg_get_tri_map: len must be obj_w*obj_h");
    if (n == 0) return HG_OK;
    HG_TRY(hg_sync(c));
    HG_TRY(run_setup(c));
    HG_TRY(ensure(c, c->d_map32, c->map32_cap, n));
    HG_TRY(ensure(c, c->d_map16, c->map16_cap, n));
    launch_map_build(mesh_of(c), frames_of(c), 0, c->pw_frames[0], c->d_map32, c->stream);
    launch_map_to_i16(c->d_map32, c->d_map16, n, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, c->d_map16, n * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HG_OK;
}

extern "C" int hg_get_tri_map_fused(hg_ctx *c, int16_t *out, size_t len)
{
    HG_TRY(bind(c));
    size_t bytes = 0;
    HG_TRY(single_frame_bytes(c, &bytes));
    const size_t n = bytes / 4;
    if (!out || len != n) return fail(c, HG_ERR_INVALID, "hg_get_tri_map_fused: len must be obj_w*obj_h");
    if (n == 0) return HG_OK;
    HG_TRY(hg_sync(c));
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, bytes));
    HG_TRY(ensure(c, c->d_map16, c->map16_cap, n));
    HG_TRY(run_setup(c, true));
    run_warp(c, c->d_out_tmp, c->d_map16);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->h_status, c->status_ptr, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->h_status[0] != FRAME_OK)
        return fail(c, HG_ERR_STATE, c->h_status[0] & FRAME_IRREGULAR ? "frame is irregular: the fused path defers it to the map path"
                                                                     : "a row overflowed the fused kernel's LDS span list");
    HIP_TRY(c, hipMemcpy(out, c->d_map16, n * sizeof(int16_t), hipMemcpyDeviceToHost));
    return HG_OK;
}

extern "C" int hg_get_matrices(hg_ctx *c, float *fwd, float *inv)
{
    HG_TRY(bind(c));
    size_t bytes = 0;
    HG_TRY(single_frame_bytes(c, &bytes));
    HG_TRY(hg_sync(c));
    if (!c->pw_setup_done) HG_TRY(run_setup(c));
    const size_t T = (size_t)c->n_tris;
    if (T == 0) return HG_OK;
    if (fwd) HIP_TRY(c, hipMemcpyAsync(fwd, c->d_fwd, sizeof(float) * 6 * T, hipMemcpyDeviceToHost, c->stream));
    std::vector<float> tmp;
    if (inv) { tmp.resize(T * kInvStride); HIP_TRY(c, hipMemcpyAsync(tmp.data(), c->d_inv, sizeof(float) * kInvStride * T, hipMemcpyDeviceToHost, c->stream)); }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (inv) for (size_t t = 0; t < T; t++) std::memcpy(inv + 6 * t, tmp.data() + kInvStride * t, sizeof(float) * 6);
    return HG_OK;
}

