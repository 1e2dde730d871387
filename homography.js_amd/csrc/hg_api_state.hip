// hg_api_state.hip -- the C ABI, part 5: the reference-state forms of the two piecewise loops.
// The reference's loops read two caches as they stand: the per-triangle matrices of the LAST setDestinyPoints (:769) and whatever
// map the shared `_trianglesCorrespondencesMatrix` field holds (:819-820 forward, :847-848 inverse; read at :957 / :1033).  A
// binding that mirrors those caches hands them over explicitly here whenever they are not the current mesh's own (the fast paths
// of hg_api_piecewise.hip / hg_api_forward.hip cover that case).  Exactness before speed: materialised map (k_tri_setup on the
// map's own point set -> k_map_fill), then the map-reading pixel loop / scatter + gather with the matrices given.
#include "hg_ctx.h"

extern "C" int hg_solve_affine_triangles(const float *src, const float *dst, int n_pts, const uint32_t *tris, int n_tris, float *out)
{
    if (!src || !dst || n_pts < 0 || n_tris < 0 || (n_tris > 0 && (!tris || !out))) return fail(nullptr, HG_ERR_INVALID, "hg_solve_affine_triangles: bad arguments");
    for (int t = 0; t < n_tris; t++) {                       // :792-800: the Float32Array scratch triangles, then affineMatrixFromTriangles
        float s[6], d[6];
        for (int k = 0; k < 3; k++) {
            const uint32_t v = tris[3 * (size_t)t + k];
            if (v < (uint32_t)n_pts) { s[2 * k] = src[2 * (size_t)v]; s[2 * k + 1] = src[2 * (size_t)v + 1]; d[2 * k] = dst[2 * (size_t)v]; d[2 * k + 1] = dst[2 * (size_t)v + 1]; }
            else s[2 * k] = s[2 * k + 1] = d[2 * k] = d[2 * k + 1] = NAN;       // typed-array read past the end: undefined -> NaN
        }
        solve_affine(s, d, out + 6 * (size_t)t);
    }
    return HG_OK;
}

static bool state_coords_ok(const float *p, size_t n)
{
    for (size_t i = 0; i < n; i++) if (std::fabs((double)p[i]) > kMaxCoord) return false;      // (NaN compares false: legal)
    return true;
}

// The Int16Array the shared field holds, as int32 cells in c->d_map32: `cells` cells in all, the first width * height of them
// rasterised from the map's own point set and triangles (fillTriangle :1111-1126 with TypedArray.fill's index rules against THAT
// length), the rest -1 (reads past the end of the reference's array give `undefined`, which fails both `> -1` and `>= 0`).
static int build_state_map(hg_ctx *c, const hg_tri_map_def *map, size_t cells)
{
    if (!map || map->n_points < 0 || map->n_triangles < 0 || (map->n_points > 0 && !map->points) || (map->n_triangles > 0 && !map->triangles))
        return fail(c, HG_ERR_INVALID, "reference-state warp: bad map definition");
    if (!state_coords_ok(map->points, (size_t)map->n_points * 2))
        return fail(c, HG_ERR_INVALID, "reference-state warp: a map coordinate is infinite or beyond 2^24 in magnitude");
    const int64_t mw = map->width, mh = map->height;
    if (mw > 0 && mh > 0 && mw * mh >= ((int64_t)1 << 31)) return fail(c, HG_ERR_INVALID, "reference-state warp: the map has 2^31 cells or more");
    if (std::abs((int64_t)map->y_off) > ((int64_t)1 << 26)) return fail(c, HG_ERR_INVALID, "reference-state warp: map y offset beyond 2^26");
    const size_t own = (mw > 0 && mh > 0) ? (size_t)(mw * mh) : 0;
    HG_TRY(ensure(c, c->d_map32, c->map32_cap, std::max(std::max(own, cells), (size_t)1)));
    if (cells > own) launch_fill_i32(c->d_map32 + own, cells - own, -1, c->stream);
    if (own == 0) { HIP_TRY(c, hipGetLastError()); return HG_OK; }
    const size_t T = (size_t)std::max(map->n_triangles, 1), N = (size_t)std::max(map->n_points, 1);
    HG_TRY(ensure(c, c->d_st_pts, c->st_pts_cap, N * 2));
    HG_TRY(ensure(c, c->d_st_tris, c->st_tris_cap, T * 3));
    HG_TRY(ensure(c, c->d_redo_frame, c->redo_frame_cap, (size_t)1));
    HG_TRY(ensure(c, c->d_redo_trir, c->redo_trir_cap, T));
    HG_TRY(ensure(c, c->d_redo_trix, c->redo_trix_cap, T));
    HG_TRY(ensure(c, c->d_redo_segs, c->redo_segs_cap, T * 3));
    HG_TRY(ensure(c, c->d_redo_fwd, c->redo_fwd_cap, T * 6));
    HG_TRY(ensure(c, c->d_redo_inv, c->redo_inv_cap, T * kInvStride));
    HG_TRY(ensure(c, c->d_redo_status, c->redo_status_cap, (size_t)1));
    FrameDesc fd; fd.x_off = 0; fd.y_off = map->y_off; fd.obj_w = map->width; fd.obj_h = map->height; fd.out_off = 0; fd.map_off = 0;
    if (map->n_points > 0) HIP_TRY(c, hipMemcpyAsync(c->d_st_pts, map->points, sizeof(float) * 2 * (size_t)map->n_points, hipMemcpyHostToDevice, c->stream));
    if (map->n_triangles > 0) HIP_TRY(c, hipMemcpyAsync(c->d_st_tris, map->triangles, sizeof(uint32_t) * 3 * (size_t)map->n_triangles, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_redo_frame, &fd, sizeof fd, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_redo_status, 0, sizeof(int32_t), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));             // (fd is local, the caller's arrays are not retained)
    PwMesh mesh = mesh_of(c);
    mesh.src_pts = c->d_st_pts; mesh.tris = c->d_st_tris; mesh.n_pts = map->n_points; mesh.n_tris = map->n_triangles;
    PwFrames fr = frames_of(c);
    fr.two_round = nullptr;                                  // (a set-up of its own, outside the frame set's step numbering)
    fr.frames = c->d_redo_frame; fr.dst_pts = c->d_st_pts; fr.trir = c->d_redo_trir; fr.trix = c->d_redo_trix; fr.band_ent = nullptr; fr.host_flag = nullptr;
    fr.segs = c->d_redo_segs; fr.fwd = c->d_redo_fwd; fr.inv = c->d_redo_inv; fr.status = c->d_redo_status; fr.n_frames = 1; fr.max_obj_h = fd.obj_h;
    if (map->n_triangles > 0) launch_tri_setup(mesh, fr, c->stream);       // edge equations + row ranges of the map's own triangles (its solves are not used)
    launch_map_build(mesh, fr, 0, fd, c->d_map32, c->stream);               // :822 / :850 fill(-1), then the rasteriser
    HIP_TRY(c, hipGetLastError());
    return HG_OK;
}

// Would a cell the loop reads name a matrix that does not exist?  (-> HG_ERR_RANGE: the reference throws at that pixel.)
static int check_state_ids(hg_ctx *c, size_t cells_read, int n_mats)
{
    HG_TRY(ensure(c, c->d_redo_status, c->redo_status_cap, (size_t)1));
    launch_map_max_i16(c->d_map32, cells_read, c->d_redo_status, c->stream);
    int32_t top = -1;
    HIP_TRY(c, hipMemcpyAsync(&top, c->d_redo_status, sizeof top, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (top >= n_mats) return fail(c, HG_ERR_RANGE, "reference-state warp: the map names triangle " + std::to_string(top) + " but only " + std::to_string(n_mats) +
                                                     " matrices exist (the reference throws a TypeError at that pixel)");
    return HG_OK;
}

static int state_common(hg_ctx *c, const float *fwd_mats, int n_mats, hg_geom geom, uint8_t *out_host, size_t *bytes)
{
    HG_TRY(bind(c));
    if (n_mats < 0 || (n_mats > 0 && !fwd_mats) || !out_host) return fail(c, HG_ERR_INVALID, "reference-state warp: bad arguments");
    if (!c->d_img) return fail(c, HG_ERR_STATE, "no source image: call hg_set_image first");
    std::vector<FrameDesc> one;
    const size_t zero = 0;
    HG_TRY(fill_frames(c, one, &geom, &zero, 1));            // (the window limits of every other entry point)
    *bytes = (geom.obj_w > 0 && geom.obj_h > 0) ? (size_t)geom.obj_w * geom.obj_h * 4 : 0;
    HG_TRY(hg_sync(c));                                      // queued runs own the scratch buffers used below
    return HG_OK;
}

extern "C" int hg_warp_inverse_piecewise_state(hg_ctx *c, const float *fwd_mats, int n_mats, const hg_tri_map_def *map, int msx, int msy,
                                               hg_geom geom, uint8_t *out_host)
{
    size_t bytes = 0;
    HG_TRY(state_common(c, fwd_mats, n_mats, geom, out_host, &bytes));
    if (!map || map->width != geom.obj_w || map->height != geom.obj_h || map->y_off != geom.y_off)
        return fail(c, HG_ERR_INVALID, "hg_warp_inverse_piecewise_state: the inverse map is the one of the output window (:1033)");
    if (bytes == 0) return HG_OK;
    const size_t cells = bytes / 4;
    HG_TRY(build_state_map(c, map, cells));
    HG_TRY(check_state_ids(c, cells, n_mats));
    // inverseAffineMatrix of every cached matrix (:1036-1038), on the host in the reference's operation order
    const size_t T = (size_t)std::max(n_mats, 1);
    std::vector<float> inv(T * kInvStride, 0.f);
    for (int t = 0; t < n_mats; t++) invert_affine(fwd_mats + 6 * (size_t)t, inv.data() + kInvStride * (size_t)t);
    HG_TRY(ensure(c, c->d_st_mats, c->st_mats_cap, T * kInvStride));
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, bytes));
    HIP_TRY(c, hipMemcpyAsync(c->d_st_mats, inv.data(), sizeof(float) * inv.size(), hipMemcpyHostToDevice, c->stream));
    PwMesh mesh = mesh_of(c);
    mesh.img = frame_img(mesh, 0); mesh.n_imgs = 1; mesh.min_src_x = msx; mesh.min_src_y = msy; mesh.n_tris = n_mats;
    PwFrames fr = frames_of(c);
    fr.two_round = nullptr;                                  // (a set-up of its own, outside the frame set's step numbering)
    fr.inv = c->d_st_mats;
    FrameDesc fd; fd.x_off = geom.x_off; fd.y_off = geom.y_off; fd.obj_w = geom.obj_w; fd.obj_h = geom.obj_h; fd.out_off = 0; fd.map_off = 0;
    launch_pw_from_map(mesh, fr, 0, fd, c->d_map32, c->d_out_tmp, c->stream);       // :1042-1056
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out_host, c->d_out_tmp, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HG_OK;
}

extern "C" int hg_warp_forward_piecewise_state(hg_ctx *c, const float *fwd_mats, int n_mats, const hg_tri_map_def *map, int msx, int msy,
                                               int max_src_x, int max_src_y, hg_geom geom, uint8_t *out_host)
{
    size_t bytes = 0;
    HG_TRY(state_common(c, fwd_mats, n_mats, geom, out_host, &bytes));
    const int64_t bw = (int64_t)max_src_x - msx, bh = (int64_t)max_src_y - msy;       // the loops :955-956
    if (bw > 0 && bh > 0) {
        if (bw * bh >= ((int64_t)1 << 31)) return fail(c, HG_ERR_INVALID, "the source-point bounding box has 2^31 pixels or more: the forward path ranks source pixels in 32 bits");
        if (bh > 65535) return fail(c, HG_ERR_INVALID, "the source-point bounding box is taller than 65535 rows: not supported by the forward path");
    }
    const size_t cells = (bw > 0 && bh > 0) ? (size_t)(bw * bh) : 0;
    HG_TRY(build_state_map(c, map, cells));
    const size_t own = (map->width > 0 && map->height > 0) ? (size_t)map->width * map->height : 0;
    HG_TRY(check_state_ids(c, std::min(own, cells), n_mats));
    // A blank output window (:440) does not stop the reference's loop: it still walks the source bounding box, reads the held map and
    // throws at a cell that names a missing matrix -- hence the check above runs first; with every id in range the loop's stores all
    // miss the empty array and nothing is left to do.
    if (bytes == 0) return HG_OK;
    const size_t T = (size_t)std::max(n_mats, 1);
    HG_TRY(ensure(c, c->d_st_mats, c->st_mats_cap, T * kInvStride));
    HG_TRY(ensure(c, c->d_out_tmp, c->out_tmp_cap, bytes));
    HG_TRY(ensure(c, c->d_win32, c->win32_cap, bytes / 4));
    if (n_mats > 0) HIP_TRY(c, hipMemcpyAsync(c->d_st_mats, fwd_mats, sizeof(float) * 6 * (size_t)n_mats, hipMemcpyHostToDevice, c->stream));
    FrameDesc fd; fd.x_off = geom.x_off; fd.y_off = geom.y_off; fd.obj_w = geom.obj_w; fd.obj_h = geom.obj_h; fd.out_off = 0; fd.map_off = 0;
    launch_fwd_pw(c->d_map32, c->d_st_mats, frame_img(mesh_of(c), 0), c->W, c->H, msx, msy, (int)std::max<int64_t>(bw, 0), (int)std::max<int64_t>(bh, 0),
                  fd, c->d_win32, c->d_out_tmp, c->stream);  // :955-969: scatter (last writer in raster order) + gather
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out_host, c->d_out_tmp, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));             // (fwd_mats / out_host are the caller's)
    return HG_OK;
}
