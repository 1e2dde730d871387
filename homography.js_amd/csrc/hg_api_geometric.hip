// hg_api_geometric.hip -- the C ABI, part 2: _inverseGeometricWarp (affine / projective), its per-frame matrix solves and frame sets.
#include "hg_ctx.h"

// ------------------------------------------------------------------------------------------------ affine / projective
static bool geo_plain_division(const double *m, const hg_geom &g) { return projective_plain_range(m, g.x_off, g.y_off, g.obj_w, g.obj_h); }   // hg_math.h

extern "C" int hg_projective_plain_range(const double *m, hg_geom geom)
{
    return (m && geo_plain_division(m, geom)) ? 1 : 0;
}

extern "C" int hg_affine_one_fma_form(const float inv[6], hg_geom geom)
{
    return (inv && affine_fusable(inv, coord_bits(geom.x_off, geom.obj_w), coord_bits(geom.y_off, geom.obj_h))) ? 1 : 0;
}

extern "C" int hg_selftest_division(hg_ctx *c, uint64_t samples, uint64_t seed, uint64_t *mismatches)
{
    HG_TRY(bind(c));
    if (!mismatches) return fail(c, HG_ERR_INVALID, "mismatches is NULL");
    unsigned long long *d = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&d), sizeof(unsigned long long)));
    *mismatches = run_selftest_division(seed, samples, d, c->stream);
    HIP_TRY(c, hipFree(d));
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

// Next staging slot with room for `bytes` (waits only if the upload that last used this slot -- eight sets ago -- is still queued).
static int geo_stage_slot(hg_ctx *c, size_t bytes, hg_ctx::GeoStage **out)
{
    const int slot = (c->geo_stage_cur + 1) % 8;
    hg_ctx::GeoStage &gs = c->geo_stage[slot];
    if (!gs.done) HIP_TRY(c, hipEventCreateWithFlags(&gs.done, hipEventDisableTiming));
    if (gs.used) HIP_TRY(c, hipEventSynchronize(gs.done));
    if (bytes > gs.cap) {
        if (gs.h) { HIP_TRY(c, hipHostFree(gs.h)); gs.h = nullptr; gs.cap = 0; }
        void *q = nullptr;
        hipError_t e = hipHostMalloc(&q, bytes + bytes / 4, hipHostMallocDefault);
        if (e != hipSuccess) return fail(c, HG_ERR_NOMEM, std::string("hipHostMalloc (frame-set staging): ") + hipGetErrorString(e));
        gs.h = static_cast<uint8_t *>(q); gs.cap = bytes + bytes / 4;
    }
    c->geo_stage_cur = slot;
    *out = &gs;
    return HG_OK;
}

extern "C" int hg_geometric_set_frames(hg_ctx *c, int kind, const double *m, const hg_geom *geoms, const size_t *offs, int n)
{
    HG_TRY(bind(c));
    if ((kind != HG_AFFINE && kind != HG_PROJECTIVE) || !m || !geoms || n <= 0) return fail(c, HG_ERR_INVALID, "hg_geometric_set_frames: bad arguments");
    // Transactional: the live frame set is dropped first and the new one only becomes visible once validation, every
    // allocation and the uploads have been queued; after a failure the next *_frames_device call returns HG_ERR_STATE
    // instead of launching F new frames against buffers sized for the old set.  No GPU wait (see hg_piecewise_set_frames).
    c->geo_frames.clear();
    std::vector<FrameDesc> fresh;
    HG_TRY(fill_frames(c, fresh, geoms, offs, n));
    HG_TRY(ensure(c, c->d_geo_frames, c->geo_frames_cap, (size_t)n));
    HG_TRY(ensure(c, c->d_mats, c->mats_cap, (size_t)n * 8));
    const size_t fd_bytes = sizeof(FrameDesc) * (size_t)n, m_bytes = sizeof(double) * 8 * (size_t)n;
    hg_ctx::GeoStage *gs = nullptr;
    HG_TRY(geo_stage_slot(c, fd_bytes + m_bytes, &gs));
    std::memcpy(gs->h, fresh.data(), fd_bytes);
    std::memcpy(gs->h + fd_bytes, m, m_bytes);
    HG_TRY(upload_staged(c, c->d_geo_frames, gs->h, fd_bytes, c->d_mats, gs->h + fd_bytes, m_bytes));
    HIP_TRY(c, hipEventRecord(gs->done, c->stream)); gs->used = true;
    c->geo_frames.swap(fresh);
    c->geo_kind = kind; c->geo_from_points = false;
    bool exact = kind == HG_AFFINE;
    for (int f = 0; f < n && exact; f++) {
        for (int k = 0; k < 6; k++) exact = exact && (double)(float)m[8 * f + k] == m[8 * f + k];
        exact = exact && std::abs((int64_t)geoms[f].x_off) + std::max(geoms[f].obj_w, 0) < (1 << 28);
    }
    if (kind == HG_PROJECTIVE) {
        exact = true;
        for (int f = 0; f < n && exact; f++) exact = geo_plain_division(m + 8 * f, geoms[f]);
    }
    c->geo_f32_exact = exact;                               // projective: "every division of the frame set is in the plain range"
    return HG_OK;
}

extern "C" int hg_geometric_set_frames_points(hg_ctx *c, int kind, const float *from, const float *to, const hg_geom *geoms, const size_t *offs, int n)
{
    HG_TRY(bind(c));
    if ((kind != HG_AFFINE && kind != HG_PROJECTIVE) || !from || !to || !geoms || n <= 0) return fail(c, HG_ERR_INVALID, "hg_geometric_set_frames_points: bad arguments");
    const size_t per = kind == HG_AFFINE ? 6 : 8;
    c->geo_frames.clear();                                  // transactional, like hg_geometric_set_frames
    std::vector<FrameDesc> fresh;
    HG_TRY(fill_frames(c, fresh, geoms, offs, n));
    HG_TRY(ensure(c, c->d_geo_frames, c->geo_frames_cap, (size_t)n));
    HG_TRY(ensure(c, c->d_mats, c->mats_cap, (size_t)n * 8));
    HG_TRY(ensure(c, c->d_geo_pts, c->geo_pts_cap, (size_t)n * 16));
    HG_TRY(ensure(c, c->d_geo_plain, c->geo_plain_cap, (size_t)n));
    const size_t fd_bytes = sizeof(FrameDesc) * (size_t)n, p_bytes = sizeof(float) * per * (size_t)n;
    hg_ctx::GeoStage *gs = nullptr;
    HG_TRY(geo_stage_slot(c, fd_bytes + 2 * p_bytes, &gs));
    std::memcpy(gs->h, fresh.data(), fd_bytes);
    std::memcpy(gs->h + fd_bytes, from, p_bytes);
    std::memcpy(gs->h + fd_bytes + p_bytes, to, p_bytes);
    HG_TRY(upload_staged(c, c->d_geo_frames, gs->h, fd_bytes, c->d_geo_pts, gs->h + fd_bytes, p_bytes,
                         c->d_geo_pts + (size_t)n * 8, gs->h + fd_bytes + p_bytes, p_bytes));
    HIP_TRY(c, hipEventRecord(gs->done, c->stream)); gs->used = true;
    c->geo_frames.swap(fresh);
    c->geo_kind = kind; c->geo_from_points = true;
    bool exact = kind == HG_AFFINE;                          // affine: the solve stores float32 values; x stays below 2^28?
    for (int f = 0; f < n && exact; f++) exact = std::abs((int64_t)geoms[f].x_off) + std::max(geoms[f].obj_w, 0) < (1 << 28);
    c->geo_f32_exact = exact;
    return HG_OK;
}

extern "C" int hg_get_geometric_matrices(hg_ctx *c, double *out, int n_frames)
{
    HG_TRY(bind(c));
    if (!out || n_frames <= 0 || (size_t)n_frames != c->geo_frames.size()) return fail(c, HG_ERR_INVALID, "hg_get_geometric_matrices: n_frames must equal the uploaded frame count");
    if (c->geo_from_points)
        launch_solve_frames(c->geo_kind, c->d_geo_pts, c->d_geo_pts + c->geo_frames.size() * 8, c->d_geo_frames, c->d_mats, c->d_geo_plain, n_frames, c->stream);
    HIP_TRY(c, hipMemcpyAsync(out, c->d_mats, sizeof(double) * 8 * n_frames, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HG_OK;
}

extern "C" int hg_warp_inverse_geometric_frames_device(hg_ctx *c, void *d_out)
{
    HG_TRY(bind(c));
    if (!d_out) return fail(c, HG_ERR_INVALID, "d_out is NULL");
    if (!c->d_img) return fail(c, HG_ERR_STATE, "no source image: call hg_set_image first");
    if (c->geo_frames.empty()) return fail(c, HG_ERR_STATE, "no frames: call hg_geometric_set_frames first");
    int mw = 0, mh = 0;
    for (const FrameDesc &d : c->geo_frames) { mw = std::max(mw, d.obj_w); mh = std::max(mh, d.obj_h); }
    if (!c->pw_pending_out.empty() || !c->fwd_pending.empty()) {     // a queued piecewise run's deferred redo must not land on these frames later
        size_t extent = 0;
        uint64_t layout = 0;
        output_layout(c->geo_frames, &extent, &layout);
        HG_TRY(settle_output_conflicts(c, d_out, extent, 0));
    }
    // the reference re-solves the inverse matrix from the swapped point sets at the head of every warp (:994): so does the step
    if (c->geo_from_points)
        launch_solve_frames(c->geo_kind, c->d_geo_pts, c->d_geo_pts + c->geo_frames.size() * 8, c->d_geo_frames, c->d_mats, c->d_geo_plain,
                            (int)c->geo_frames.size(), c->stream);
    HG_TRY(time_begin(c));
    launch_geo(c->geo_kind, c->geo_f32_exact, c->d_geo_frames, c->d_mats, (int)c->geo_frames.size(), mw, mh, c->d_img, c->W, c->H,
               c->n_imgs, (uint64_t)c->img_stride, static_cast<uint8_t *>(d_out),
               (c->geo_from_points && c->geo_kind == HG_PROJECTIVE) ? c->d_geo_plain : nullptr, c->opt_geo_nw,
               c->xcc_log2, c->opt_xcc_rotate >= 0 ? c->opt_xcc_rotate != 0 : c->n_imgs > 1,   // (XCD bands; rotating with the frame when every frame reads its own source)
               c->stream);
    HG_TRY(time_end(c));
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

extern "C" int hg_warp_inverse_geometric_batch_device(hg_ctx *c, int kind, const double *m, const hg_geom *geoms,
                                                      const size_t *offs, int n, void *d_out)
{
    HG_TRY(hg_geometric_set_frames(c, kind, m, geoms, offs, n));
    return hg_warp_inverse_geometric_frames_device(c, d_out);
}

extern "C" int hg_warp_inverse_geometric_device(hg_ctx *c, int kind, const double *m, hg_geom geom, void *d_out)
{
    if (!m) return fail(c, HG_ERR_INVALID, "m is NULL");
    double m8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    std::memcpy(m8, m, sizeof(double) * (kind == HG_AFFINE ? 6 : 8));
    const size_t zero = 0;
    return hg_warp_inverse_geometric_batch_device(c, kind, m8, &geom, &zero, 1, d_out);
}

extern "C" int hg_warp_inverse_geometric(hg_ctx *c, int kind, const double *m, hg_geom geom, uint8_t *out_host)
{
    HG_TRY(bind(c));
    if (!out_host) return fail(c, HG_ERR_INVALID, "out is NULL");
    if (geom.obj_w <= 0 || geom.obj_h <= 0) return HG_OK;
    const size_t bytes = (size_t)geom.obj_w * geom.obj_h * 4;
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, bytes));
    HG_TRY(hg_warp_inverse_geometric_device(c, kind, m, geom, c->d_out_tmp));
    HIP_TRY(c, hipMemcpyAsync(out_host, c->d_out_tmp, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HG_OK;
}

