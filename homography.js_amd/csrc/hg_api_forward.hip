// hg_api_forward.hip -- the C ABI, part 4: the forward (scatter-semantics) warps _geometricWarp / _piecewiseAffineWarp.
#include "hg_ctx.h"

// ------------------------------------------------------------------------------------------------ forward (scatter) paths
// Limits shared by the forward (scatter) entry points: the raster rank of a source pixel is an int32 (y*W + x resp. the
// cell of the source-bbox map), the scatter kernels put source rows in grid.y, and every window passes the same checks as the
// inverse paths (fill_frames: 2^31 pixels, offsets within 2^26, 4-byte aligned output offsets).
static int forward_limits(hg_ctx *c, int64_t w, int64_t h, const char *what)
{
    if (w <= 0 || h <= 0) return HG_OK;
    if (w * h >= ((int64_t)1 << 31)) return fail(c, HG_ERR_INVALID, std::string(what) + " has 2^31 pixels or more: the forward path ranks source pixels in 32 bits");
    if (h > 65535) return fail(c, HG_ERR_INVALID, std::string(what) + " is taller than 65535 rows: not supported by the forward path");
    return HG_OK;
}

// Is k_fwd_tiles admissible for this frame (see the kernel: every bound below keeps the rounding error of its candidate
// constraints under the 1/64-pixel widening, and no source pixel further than kFwdWrap columns outside the window)?
// Fills P (matrix, inverse = adjugate / det of the 3x3 form, use_inv only when the inverse reproduces the source corners).
static bool fwd_tile_param(int kind, const double *m, int W, int H, const FrameDesc &fd, FwdParam &P)
{
    if (fd.obj_w < 2 * kFwdWrap || fd.obj_h <= 0 || W <= 0 || H <= 0 || W > 65535 || H > 65535) return false;
    const bool proj = kind == HG_PROJECTIVE;
    for (int k = 0; k < (proj ? 8 : 6); k++) if (!std::isfinite(m[k])) return false;
    const double lim = proj ? 1.0e4 : 1.0e6, flim = proj ? 1.0e5 : 1.0e7;
    for (int k = 0; k < 6; k++) if (std::fabs(m[k]) > lim) return false;
    if (proj && (std::fabs(m[6]) > 0.1 || std::fabs(m[7]) > 0.1)) return false;
    double Hm[9];
    if (proj) { for (int k = 0; k < 8; k++) Hm[k] = m[k]; Hm[8] = 1.0; }
    else { Hm[0] = m[0]; Hm[1] = m[2]; Hm[2] = m[4]; Hm[3] = m[1]; Hm[4] = m[3]; Hm[5] = m[5]; Hm[6] = 0.0; Hm[7] = 0.0; Hm[8] = 1.0; }
    double umin = INFINITY, umax = -INFINITY, cfx[4], cfy[4];
    for (int c = 0; c < 4; c++) {
        const double x = (c & 1) ? W - 1 : 0, y = (c & 2) ? H - 1 : 0;
        const double den = Hm[6] * x + Hm[7] * y + Hm[8];
        if (!(den >= 1.0e-2)) return false;                  // (linear: positive at the corners = positive on the whole image)
        const double fx = (Hm[0] * x + Hm[1] * y + Hm[2]) / den, fy = (Hm[3] * x + Hm[4] * y + Hm[5]) / den;
        if (!(std::fabs(fx) < flim && std::fabs(fy) < flim)) return false;
        cfx[c] = fx; cfy[c] = fy;
        umin = std::min(umin, fx - fd.x_off); umax = std::max(umax, fx - fd.x_off);
    }
    // the image of the source rectangle is the convex hull of its corner images: no rounded x further out than kFwdWrap - 2
    if (umin < -(double)(kFwdWrap - 2) || umax > (double)fd.obj_w + (kFwdWrap - 2)) return false;
    for (int k = 0; k < 8; k++) P.m[k] = k < (proj ? 8 : 6) ? m[k] : 0.0;
    P.use_inv = 0; P.pad = 0;
    const double det = Hm[0] * (Hm[4] * Hm[8] - Hm[5] * Hm[7]) - Hm[1] * (Hm[3] * Hm[8] - Hm[5] * Hm[6]) + Hm[2] * (Hm[3] * Hm[7] - Hm[4] * Hm[6]);
    const double adj[9] = { Hm[4] * Hm[8] - Hm[5] * Hm[7], Hm[2] * Hm[7] - Hm[1] * Hm[8], Hm[1] * Hm[5] - Hm[2] * Hm[4],
                            Hm[5] * Hm[6] - Hm[3] * Hm[8], Hm[0] * Hm[8] - Hm[2] * Hm[6], Hm[2] * Hm[3] - Hm[0] * Hm[5],
                            Hm[3] * Hm[7] - Hm[4] * Hm[6], Hm[1] * Hm[6] - Hm[0] * Hm[7], Hm[0] * Hm[4] - Hm[1] * Hm[3] };
    for (int k = 0; k < 9; k++) P.inv[k] = 0.0;
    if (std::isfinite(det) && std::fabs(det) > 1.0e-12) {
        bool ok = true;
        for (int k = 0; k < 9; k++) { P.inv[k] = adj[k] / det; if (!std::isfinite(P.inv[k])) ok = false; }
        for (int c = 0; c < 4 && ok; c++) {                  // the inverse has to bring the corner images back (1e-3 px), with w > 0
            const double x = (c & 1) ? W - 1 : 0, y = (c & 2) ? H - 1 : 0;
            const double X = P.inv[0] * cfx[c] + P.inv[1] * cfy[c] + P.inv[2], Y = P.inv[3] * cfx[c] + P.inv[4] * cfy[c] + P.inv[5];
            const double Wd = P.inv[6] * cfx[c] + P.inv[7] * cfy[c] + P.inv[8];
            if (!(Wd > 0.0) || !(std::fabs(X / Wd - x) < 1.0e-3) || !(std::fabs(Y / Wd - y) < 1.0e-3)) ok = false;
        }
        P.use_inv = ok ? 1 : 0;
    }
    return true;
}

extern "C" int hg_forward_tiles_admissible(int kind, const double *m, int W, int H, hg_geom geom)
{
    if ((kind != HG_AFFINE && kind != HG_PROJECTIVE) || !m) return 0;
    FrameDesc fd; fd.x_off = geom.x_off; fd.y_off = geom.y_off; fd.obj_w = geom.obj_w; fd.obj_h = geom.obj_h; fd.out_off = 0; fd.map_off = 0;
    FwdParam P;
    if (!fwd_tile_param(kind, m, W, H, fd, P)) return 0;
    return P.use_inv ? 2 : 1;
}

extern "C" int hg_warp_forward_geometric_batch_device(hg_ctx *c, int kind, const double *m, const hg_geom *geoms, const size_t *offs, int n, void *d_out)
{
    HG_TRY(bind(c));
    if ((kind != HG_AFFINE && kind != HG_PROJECTIVE) || !m || !geoms || n <= 0 || !d_out) return fail(c, HG_ERR_INVALID, "hg_warp_forward_geometric: bad arguments");
    if (!c->d_img) return fail(c, HG_ERR_STATE, "no source image: call hg_set_image first");
    HG_TRY(forward_limits(c, c->W, c->H, "the source image"));
    std::vector<FrameDesc> fds;
    HG_TRY(fill_frames(c, fds, geoms, offs, n));
    size_t max_px = 0;
    for (const FrameDesc &fd : fds) if (fd.obj_w > 0 && fd.obj_h > 0) max_px = std::max(max_px, (size_t)fd.obj_w * fd.obj_h);
    if (max_px == 0) return HG_OK;
    HG_TRY(hg_sync(c));
    // Tile-binned gather (k_fwd_tiles, all frames in one launch) when every frame is admissible and the windows are large
    // enough to fill the chip with tiles of a bounded candidate count; otherwise scatter + gather frame after frame.
    if (c->opt_fwd_tiles != 0) {
        std::vector<FwdParam> par((size_t)n);
        bool all = true;
        int64_t tiles = 0;
        int mw = 0, mh = 0;
        for (int f = 0; f < n && all; f++) {
            if (fds[f].obj_w <= 0 || fds[f].obj_h <= 0) { std::memset(&par[f], 0, sizeof(FwdParam)); continue; }
            all = fwd_tile_param(kind, m + 8 * (size_t)f, c->W, c->H, fds[f], par[f]);
            const int64_t t = (int64_t)((fds[f].obj_w + kFwdTileW - 1) / kFwdTileW) * ((fds[f].obj_h + kFwdTileH - 1) / kFwdTileH);
            if (c->opt_fwd_tiles < 0 && (int64_t)c->W * c->H > t * 32768) all = false;   // (few tiles for many source pixels: long serial candidate loops)
            tiles += t; mw = std::max(mw, fds[f].obj_w); mh = std::max(mh, fds[f].obj_h);
        }
        if (all && c->opt_fwd_tiles < 0 && tiles < 160) all = false;     // (measured break-even against the three small scatter-path kernels)
        if (all) {
            FwdBatch batch;
            std::memset(&batch, 0, sizeof batch);
            if (n == 1) { batch.p0 = par[0]; batch.f0 = fds[0]; }
            else {
                const size_t bytes = sizeof(FwdParam) * n + sizeof(FrameDesc) * n;
                HG_TRY(ensure(c, c->d_fwd_par, c->fwd_par_cap, bytes));
                std::vector<uint8_t> blob(bytes);
                std::memcpy(blob.data(), par.data(), sizeof(FwdParam) * n);
                std::memcpy(blob.data() + sizeof(FwdParam) * n, fds.data(), sizeof(FrameDesc) * n);
                HIP_TRY(c, hipMemcpyAsync(c->d_fwd_par, blob.data(), bytes, hipMemcpyHostToDevice, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream)); // caller / local memory is not retained
                batch.params = reinterpret_cast<const FwdParam *>(c->d_fwd_par);
                batch.frames = reinterpret_cast<const FrameDesc *>(c->d_fwd_par + sizeof(FwdParam) * n);
            }
            launch_fwd_tiles(kind, batch, n, mw, mh, c->d_img, c->n_imgs, c->img_stride, c->W, c->H, static_cast<uint8_t *>(d_out), c->stream);
            HIP_TRY(c, hipGetLastError());
            c->fwd_last_kernel = 2;
            return HG_OK;
        }
    }
    c->fwd_last_kernel = 1;
    HG_TRY(ensure(c, c->d_mats, c->mats_cap, (size_t)8 * n));
    HG_TRY(ensure(c, c->d_win32, c->win32_cap, max_px));
    HIP_TRY(c, hipMemcpyAsync(c->d_mats, m, sizeof(double) * 8 * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));         // caller memory is not retained
    c->geo_frames.clear();                               // the uploaded geometric frame set was overwritten
    for (int f = 0; f < n; f++)                          // frames run back to back on the stream (one winner buffer, reused in order)
        launch_fwd_geo(kind, c->d_mats + 8 * (size_t)f, frame_img(mesh_of(c), f), c->W, c->H, fds[f], c->d_win32, static_cast<uint8_t *>(d_out), c->stream);
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

extern "C" int hg_warp_forward_geometric_device(hg_ctx *c, int kind, const double *m, hg_geom geom, void *d_out)
{
    if (!m) return fail(c, HG_ERR_INVALID, "m is NULL");
    double m8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    std::memcpy(m8, m, sizeof(double) * (kind == HG_AFFINE ? 6 : 8));
    const size_t zero = 0;
    return hg_warp_forward_geometric_batch_device(c, kind, m8, &geom, &zero, 1, d_out);
}

extern "C" int hg_warp_forward_geometric(hg_ctx *c, int kind, const double *m, hg_geom geom, uint8_t *out_host)
{
    HG_TRY(bind(c));
    if (!out_host) return fail(c, HG_ERR_INVALID, "hg_warp_forward_geometric: bad arguments");
    if (geom.obj_w <= 0 || geom.obj_h <= 0) return HG_OK;
    const size_t n = (size_t)geom.obj_w * geom.obj_h;
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, n * 4));
    HG_TRY(hg_warp_forward_geometric_device(c, kind, m, geom, c->d_out_tmp));
    HIP_TRY(c, hipMemcpyAsync(out_host, c->d_out_tmp, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HG_OK;
}

// Per-row extents of every matrix index inside its bbox (k_fwd_pw_tiles' candidate bound), built once per forward map when the
// tile path first wants them: offsets are a prefix sum of the bbox heights (host).
static int ensure_fwd_rowext(hg_ctx *c, int map_w, int map_h)
{
    if (c->fwd_rowext_ok || c->n_tris <= 0) return HG_OK;
    std::vector<int32_t> bb((size_t)4 * c->n_tris);
    HIP_TRY(c, hipMemcpyAsync(bb.data(), c->d_fbbox, sizeof(int32_t) * bb.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<uint32_t> off((size_t)c->n_tris);
    uint64_t total = 0;
    for (int t = 0; t < c->n_tris; t++) {
        off[t] = (uint32_t)total;
        if (bb[4 * (size_t)t + 3] >= bb[4 * (size_t)t + 1]) total += (uint64_t)(bb[4 * (size_t)t + 3] - bb[4 * (size_t)t + 1] + 1);
        if (total > ((uint64_t)1 << 27)) { c->fwd_pw_tiles_disabled = true; return HG_OK; }   // fill() wrap-around quirks made the boxes absurdly tall: scatter path
    }
    HG_TRY(ensure(c, c->d_frowoff, c->frowoff_cap, (size_t)c->n_tris));
    HG_TRY(ensure(c, c->d_frowext, c->frowext_cap, (size_t)2 * std::max<uint64_t>(total, 1)));
    HIP_TRY(c, hipMemcpyAsync(c->d_frowoff, off.data(), sizeof(uint32_t) * off.size(), hipMemcpyHostToDevice, c->stream));
    launch_fmap_rowext(c->d_fmap, map_w, map_h, c->d_fbbox, c->d_frowoff, c->d_frowext, (size_t)total, c->n_tris, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));             // (`off` is local)
    c->fwd_rowext_ok = true;
    return HG_OK;
}

// _piecewiseAffineWarp :948-972 for n destination point sets on the current mesh (the caller loop `setDestinyPoints(d_f); warp()`
// when warp() takes the forward path, :421), asynchronous, frames in GPU memory.
extern "C" int hg_warp_forward_piecewise_batch_device(hg_ctx *c, const float *dst_points, int max_src_x, int max_src_y, const hg_geom *geoms,
                                                      const size_t *offs, int n, void *d_out)
{
    HG_TRY(bind(c));
    if (!dst_points || !geoms || n <= 0 || !d_out) return fail(c, HG_ERR_INVALID, "hg_warp_forward_piecewise: bad arguments");
    if (!c->d_img) return fail(c, HG_ERR_STATE, "no source image: call hg_set_image first");
    if (!c->have_mesh) return fail(c, HG_ERR_STATE, "no mesh: call hg_piecewise_set_mesh first");
    const int64_t map_w = (int64_t)max_src_x - c->min_src_x, map_h = (int64_t)max_src_y - c->min_src_y;
    HG_TRY(forward_limits(c, map_w, map_h, "the source-point bounding box"));
    const size_t n_map = (map_w > 0 && map_h > 0) ? (size_t)map_w * map_h : 0;
    // (A) forward triangle map over the source bbox: _buildTrianglesCorrespondencesMatrix :817-832 == the same
    //     rasteriser on the SOURCE triangles with width maxSrcX-minSrcX and y offset minSrcY.  It depends on the mesh only,
    //     so it is kept until the mesh (or the bbox) changes -- like the reference's cached _trianglesCorrespondencesMatrix.
    hg_geom gmap = { 0, c->min_src_y, (int32_t)map_w, (int32_t)map_h };
    const size_t zero = 0;
    if (!c->pw_pending_out.empty()) HG_TRY(hg_sync(c));       // (queued inverse runs: one order of deferred redos per output buffer)
    if (n_map && !(c->fmap_valid && c->fmap_w == map_w && c->fmap_h == map_h)) {
        if (!c->fwd_pending.empty()) HG_TRY(hg_sync(c));     // (queued batches are redone over the map they ran on)
        HG_TRY(hg_piecewise_set_frames(c, c->h_src.data(), &gmap, &zero, 1));
        c->status_ptr = c->d_status;
        HIP_TRY(c, hipMemsetAsync(c->d_status, 0, sizeof(int32_t), c->stream));
        { PwFrames fr_ = frames_of(c); fr_.band_ent = nullptr; fr_.two_round = nullptr; launch_tri_setup(mesh_of(c), fr_, c->stream); }   // (no candidate bands: the forward kernels have their own tile lists)
        HG_TRY(ensure(c, c->d_fmap, c->fmap_cap, n_map));
        launch_map_build(mesh_of(c), frames_of(c), 0, c->pw_frames[0], c->d_fmap, c->stream);
        HG_TRY(ensure(c, c->d_fbbox, c->fbbox_cap, (size_t)4 * std::max(c->n_tris, 1)));
        launch_fmap_bbox(c->d_fmap, (int)map_w, (int)map_h, c->d_fbbox, c->n_tris, c->stream);     // (for the tile-binned frames below)
        HIP_TRY(c, hipGetLastError());
        c->fmap_valid = true; c->fmap_w = (int)map_w; c->fmap_h = (int)map_h;
        c->fwd_rowext_ok = false;
    }
    // (B) forward matrices of every frame (:785-804) in one launch, then scatter + gather frame after frame
    c->pw_quick_layout = true;                               // (the inverse kernels' layout estimate is not needed here: no host walk over the triangles)
    const int rc_frames = hg_piecewise_set_frames(c, dst_points, geoms, offs, n);
    c->pw_quick_layout = false;
    HG_TRY(rc_frames);
    c->status_ptr = c->d_status;                             // (k_tri_setup only ORs flags into these words and nothing on the forward path reads them: not cleared)
    { PwFrames fr_ = frames_of(c); fr_.band_ent = nullptr; fr_.two_round = nullptr; launch_tri_setup(mesh_of(c), fr_, c->stream); }   // (no candidate bands: the forward kernels have their own tile lists)
    c->pw_setup_done = false;
    size_t out_extent = 0;
    uint64_t out_layout = 0;
    output_layout(c->pw_frames, &out_extent, &out_layout);
    size_t max_px = 0;
    for (const FrameDesc &fd : c->pw_frames) if (fd.obj_w > 0 && fd.obj_h > 0) max_px = std::max(max_px, (size_t)fd.obj_w * fd.obj_h);
    if (max_px) {
        // Tile-binned gather (k_fwd_pw_bins + k_fwd_pw_tiles, all frames in two launches) when the batch has enough tiles;
        // frames the device cannot bound (flagged in their status word) are redone through scatter + gather by hg_sync.
        int mw = 0, mh = 0;
        int64_t tiles = 0;
        for (const FrameDesc &fd : c->pw_frames) if (fd.obj_w > 0 && fd.obj_h > 0) {
            mw = std::max(mw, fd.obj_w); mh = std::max(mh, fd.obj_h);
            tiles += (int64_t)((fd.obj_w + kFwdTileW - 1) / kFwdTileW) * ((fd.obj_h + kFwdTileH - 1) / kFwdTileH);
        }
        const int tsx = (mw + kFwdTileW - 1) / kFwdTileW, tsy = (mh + kFwdTileH - 1) / kFwdTileH;
        bool use_tiles = n_map > 0 && c->n_tris > 0 && !c->fwd_pw_tiles_disabled && tsy <= 65535 && map_w <= 65535 &&   // (winner keys: 16 bits per map coordinate)
                         (c->opt_fwd_tiles > 0 || (c->opt_fwd_tiles < 0 && tiles >= 160 && (int64_t)n_map <= tiles * 32768 &&
                                                   (int64_t)c->n_tris * n <= 8 * tiles));   // (measured up to 6 triangles per tile, 4K 96 x 54 cells: 171 -> 92 us per frame; denser: not measured, scatter path)
        if (use_tiles) { HG_TRY(ensure_fwd_rowext(c, (int)map_w, (int)map_h)); use_tiles = c->fwd_rowext_ok; }
        if (use_tiles) {
            // The tile counters and the per-frame status words are ZERO between calls: k_fwd_pw_tiles clears the counter of every
            // tile it consumes, hg_sync the status words it found set -- no memset in front of every batch (two stream operations
            // less per call: they were a fifth of a single 4K frame's time).  Status words: a ring of sets, one per queued batch.
            const size_t per = (size_t)n * tsx * tsy;
            { const size_t cap0 = c->ftile_cnt_cap;
              HG_TRY(ensure(c, c->d_ftile_cnt, c->ftile_cnt_cap, per));
              if (c->ftile_cnt_cap != cap0) HIP_TRY(c, hipMemsetAsync(c->d_ftile_cnt, 0, sizeof(int32_t) * c->ftile_cnt_cap, c->stream)); }
            if (c->fwd_pending.size() >= kFwdStatusRing - 1 || (size_t)n > c->fwd_status_stride) {
                HG_TRY(hg_sync(c));                              // ring full, or a larger batch than the ring's sets were laid out for
                if ((size_t)n > c->fwd_status_stride) {
                    HG_TRY(ensure(c, c->d_fwd_status, c->fwd_status_cap, (size_t)n * kFwdStatusRing));
                    c->fwd_status_stride = c->fwd_status_cap / kFwdStatusRing;
                    HIP_TRY(c, hipMemsetAsync(c->d_fwd_status, 0, sizeof(int32_t) * c->fwd_status_cap, c->stream));
                }
            }
            c->fwd_slot = (c->fwd_slot + 1) % (int)kFwdStatusRing;
            HG_TRY(ensure(c, c->d_ftile_ent, c->ftile_ent_cap, per * (size_t)c->fwd_pw_cap));
            FwdPwTiles p;
            p.fmap = c->d_fmap; p.fwd = c->d_fwd; p.bbox = c->d_fbbox; p.frames = c->d_pw_frames; p.rowext = c->d_frowext; p.rowoff = c->d_frowoff;
            p.tile_cnt = c->d_ftile_cnt; p.tile_ent = c->d_ftile_ent; p.status = c->d_fwd_status + (size_t)c->fwd_slot * c->fwd_status_stride;
            p.host_flag = c->h_flag ? c->h_flag + 1 : nullptr;
            p.T = c->n_tris; p.min_src_x = c->min_src_x; p.min_src_y = c->min_src_y; p.map_w = (int)map_w; p.map_h = (int)map_h;
            p.tsx = tsx; p.tsy = tsy; p.cap = c->fwd_pw_cap;
            launch_fwd_pw_tiles(p, n, mw, mh, c->d_img, c->n_imgs, c->img_stride, c->W, c->H, static_cast<uint8_t *>(d_out), c->stream);
            HIP_TRY(c, hipGetLastError());
            { hg_ctx::FwdPending fp;
              fp.out = static_cast<uint8_t *>(d_out); fp.n = n; fp.slot = c->fwd_slot; fp.stage = c->stage_cur; fp.max_src_x = max_src_x; fp.max_src_y = max_src_y;
              fp.extent = out_extent; fp.layout = out_layout;
              c->fwd_pending.push_back(fp); }
            c->fwd_last_kernel = 2;
            return HG_OK;
        }
        c->fwd_last_kernel = 1;
        HG_TRY(settle_output_conflicts(c, d_out, out_extent, 0));      // (this path keeps no pending record: nothing queued may be redone over it later)
        HG_TRY(ensure(c, c->d_win32, c->win32_cap, max_px));
        for (int f = 0; f < n; f++)
            launch_fwd_pw(c->d_fmap, c->d_fwd + (size_t)f * c->n_tris * 6, frame_img(mesh_of(c), f), c->W, c->H, c->min_src_x, c->min_src_y, (int)map_w, (int)map_h,
                          c->pw_frames[f], c->d_win32, static_cast<uint8_t *>(d_out), c->stream);
    }
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

extern "C" int hg_warp_forward_piecewise_device(hg_ctx *c, const float *dst_points, int max_src_x, int max_src_y, hg_geom geom, void *d_out)
{
    const size_t zero = 0;
    return hg_warp_forward_piecewise_batch_device(c, dst_points, max_src_x, max_src_y, &geom, &zero, 1, d_out);
}

extern "C" int hg_warp_forward_piecewise(hg_ctx *c, const float *dst_points, int max_src_x, int max_src_y, hg_geom geom, uint8_t *out_host)
{
    HG_TRY(bind(c));
    if (!out_host) return fail(c, HG_ERR_INVALID, "hg_warp_forward_piecewise: bad arguments");
    if (geom.obj_w <= 0 || geom.obj_h <= 0) return HG_OK;
    const size_t n = (size_t)geom.obj_w * geom.obj_h;
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, n * 4));
    HG_TRY(hg_warp_forward_piecewise_device(c, dst_points, max_src_x, max_src_y, geom, c->d_out_tmp));
    HG_TRY(hg_sync(c));                                      // (settles a frame the tile kernels handed to the scatter path)
    HIP_TRY(c, hipMemcpyAsync(out_host, c->d_out_tmp, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HG_OK;
}

