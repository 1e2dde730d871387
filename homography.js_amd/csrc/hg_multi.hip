// hg_multi.hip -- several GPUs of one node behind ONE host thread, on top of the public C ABI (include/hgwarp.h).
//
// What it replaces: the caller loop `for (f) { setDestinyPoints(dst_f); warp(); }` (test/benchmark.js:107-110) when the
// host wants the frames of that loop spread over G devices: frames are independent units, so each device
// gets a contiguous block of them and there is no data-path collective; the only exchange is the one-off fan-out of the
// shared source texture.  This is the path for hosts without torch.distributed / RCCL process groups (the Node binding:
// one process, one thread); bench.py's one-process-per-GPU launch uses homography.js_amd/dist.py over RCCL instead.
//
// Source fan-out over xGMI (point-to-point links, 7 x ~153 GB/s per GPU; a ring or a chain is bound by ONE link):
//   1. H2D of the image into device 0;
//   2. scatter: device 0 -> peer p, slice p (1/G of the image) -- every link 0->p carries 1/G;
//   3. all-gather: every device sends the slice it owns to every other device -- every link p->q carries 1/G;
// all as hipMemcpyPeerAsync on per-device streams ordered by events.  The plan is made PER PAIR (hg_multi_plan_fanout, a pure
// function with a CPU test): where hipDeviceCanAccessPeer is 0 for a pair, that one copy falls back -- a slice whose owner the
// reader cannot reach comes from the root instead, and from the caller's host buffer if the root is out of reach too; the pairs
// without access are listed by hg_multi_peer_note.
#include "../../include/hgwarp.h"
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

struct hg_multi {
    struct Dev {
        int id = 0;
        hg_ctx *ctx = nullptr;
        hipStream_t copy = nullptr;              // fan-out / gather stream of this device
        hipEvent_t have_slice = nullptr;         // this device's own slice of the source has arrived
        hipEvent_t have_image = nullptr;         // the whole source has arrived on this device (its warps wait for this, the host does not)
        uint8_t *d_img = nullptr; size_t img_cap = 0;
        uint8_t *d_imgs = nullptr; size_t imgs_cap = 0;   // per-frame sources of this device's block (hg_multi_*_images)
        uint8_t *d_out = nullptr; size_t out_cap = 0;
        int first = 0, count = 0;                // frames of the current batch
        std::vector<size_t> offs;                // their byte offsets in d_out
    };
    std::vector<Dev> devs;
    std::vector<uint8_t> access;                 // access[p * G + q] = 1: device q copies from device p's memory directly (peer access enabled, or the same GPU)
    std::string peer_note;                       // "" or the pairs without peer access, for diagnostics
    int W = 0, H = 0;
    int n_pts = 0;
    std::vector<hg_geom> geoms;                  // of the current batch
    std::string err;
};

static thread_local std::string g_merr;

static int mfail(hg_multi *m, int code, const std::string &msg)
{
    if (m) m->err = msg;
    g_merr = msg;
    return code;
}

#define MHIP(m, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return mfail((m), HG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define MHG(m, ctx, expr) do { int s_ = (expr); if (s_ != HG_OK) return mfail((m), s_, hg_last_error(ctx)); } while (0)

extern "C" int hg_multi_partition(int n_frames, int n_devices, int index, int *first, int *count)
{
    if (n_frames < 0 || n_devices <= 0 || index < 0 || index >= n_devices || !first || !count) return mfail(nullptr, HG_ERR_INVALID, "hg_multi_partition: bad arguments");
    const int base = n_frames / n_devices, extra = n_frames % n_devices;
    *first = index * base + (index < extra ? index : extra);
    *count = base + (index < extra ? 1 : 0);
    return HG_OK;
}

extern "C" const char *hg_multi_last_error(const hg_multi *m) { return m ? m->err.c_str() : g_merr.c_str(); }
extern "C" int hg_multi_device_count(const hg_multi *m) { return m ? (int)m->devs.size() : 0; }
extern "C" hg_ctx *hg_multi_ctx(hg_multi *m, int i) { return (m && i >= 0 && i < (int)m->devs.size()) ? m->devs[i].ctx : nullptr; }

extern "C" void hg_multi_destroy(hg_multi *m)
{
    if (!m) return;
    for (auto &d : m->devs) {
        if (d.ctx) {
            (void)hg_sync(d.ctx);
            (void)hipSetDevice(d.id);
            if (d.copy) { (void)hipStreamSynchronize(d.copy); (void)hipStreamDestroy(d.copy); }
            if (d.have_slice) (void)hipEventDestroy(d.have_slice);
            if (d.have_image) (void)hipEventDestroy(d.have_image);
            hg_destroy(d.ctx);                   // (drops its alias of d_img first)
            if (d.d_img) (void)hipFree(d.d_img);
            if (d.d_imgs) (void)hipFree(d.d_imgs);
            if (d.d_out) (void)hipFree(d.d_out);
        }
    }
    delete m;
}

extern "C" int hg_multi_create(const int *device_ids, int n_devices, hg_multi **out)
{
    if (!out) return mfail(nullptr, HG_ERR_INVALID, "hg_multi_create: out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices <= 0 || n_devices > 64) return mfail(nullptr, HG_ERR_INVALID, "hg_multi_create: need 1..64 device ids");
    hg_multi *m = new hg_multi();
    m->devs.resize(n_devices);
    for (int i = 0; i < n_devices; i++) {
        auto &d = m->devs[i];
        d.id = device_ids[i];
        int rc = hg_create(d.id, &d.ctx);
        if (rc != HG_OK) { const std::string why = hg_last_error(nullptr); hg_multi_destroy(m); return mfail(nullptr, rc, why); }
        if (hipSetDevice(d.id) != hipSuccess || hipStreamCreateWithFlags(&d.copy, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&d.have_slice, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&d.have_image, hipEventDisableTiming) != hipSuccess) {
            hg_multi_destroy(m);
            return mfail(nullptr, HG_ERR_HIP, "hg_multi_create: stream / event creation failed");
        }
    }
    // peer access, pair by pair (a device listed twice talks to itself: plain device copies).  A pair without it does not switch the
    // whole fan-out to per-device H2D copies any more: only the copies of that pair take another route (hg_multi_plan_fanout).
    m->access.assign((size_t)n_devices * n_devices, 1);
    for (int i = 0; i < n_devices; i++)
        for (int j = 0; j < n_devices; j++) {
            const int a = m->devs[i].id, b = m->devs[j].id;
            if (a == b) continue;
            int can = 0;
            bool ok = hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can;
            if (ok) {
                (void)hipSetDevice(a);
                const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
                ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
                (void)hipGetLastError();
            }
            if (!ok) {                                       // device a cannot map device b's memory: copies b -> a take another route
                m->access[(size_t)j * n_devices + i] = 0;
                m->peer_note += (m->peer_note.empty() ? "no peer access: " : ", ") + std::to_string(b) + "->" + std::to_string(a);
            }
        }
    *out = m;
    return HG_OK;
}

static int ensure_dev(hg_multi *m, hg_multi::Dev &d, uint8_t *&p, size_t &cap, size_t need)
{
    if (need <= cap) return HG_OK;
    MHIP(m, hipSetDevice(d.id));
    if (p) { MHG(m, d.ctx, hg_sync(d.ctx)); MHIP(m, hipFree(p)); p = nullptr; cap = 0; }
    void *q = nullptr;
    if (hipMalloc(&q, need) != hipSuccess) return mfail(m, HG_ERR_NOMEM, "hg_multi: hipMalloc failed");
    p = static_cast<uint8_t *>(q); cap = need;
    return HG_OK;
}

// ---- the fan-out plan (pure host logic; tests/test_cabi_cpu.py runs it without a GPU)
// access[p * G + q] != 0: device q can copy straight out of device p's memory.  Ops in issue order, 4 ints each:
//   {dst device index, src device index or -1 = the caller's host buffer, slice, phase}   phase 0 = scatter, 1 = all-gather.
// Scatter: slice q -> device q from the root (device 0), else from the host.  All-gather: device q takes slice p from its owner p
// (staggered, so that the pulls of one step use different links), else from the root -- which holds every slice --, else from the
// host.  With full access every ordered pair p -> q carries exactly one slice (1/G of the image per link).
struct FanOp { int dst, src, slice, phase; };
static void plan_fanout(int G, const uint8_t *access, std::vector<FanOp> &ops)
{
    auto can = [&](int p, int q) { return access[(size_t)p * G + q] != 0; };
    ops.clear();
    for (int q = 1; q < G; q++) ops.push_back({q, can(0, q) ? 0 : -1, q, 0});
    for (int q = 1; q < G; q++)
        for (int k = 1; k < G; k++) {
            const int p = (q + k) % G;
            if (p == q) continue;
            ops.push_back({q, can(p, q) ? p : (can(0, q) ? 0 : -1), p, 1});
        }
}

extern "C" int hg_multi_plan_fanout(int n_devices, const uint8_t *access, int32_t *ops, int max_ops)
{
    if (n_devices <= 0 || n_devices > 64 || !access) return -1;
    std::vector<FanOp> plan;
    plan_fanout(n_devices, access, plan);
    if (ops) for (size_t i = 0; i < plan.size() && (int)i < max_ops; i++) { ops[4 * i] = plan[i].dst; ops[4 * i + 1] = plan[i].src; ops[4 * i + 2] = plan[i].slice; ops[4 * i + 3] = plan[i].phase; }
    return (int)plan.size();
}

extern "C" const char *hg_multi_peer_note(const hg_multi *m) { return m ? m->peer_note.c_str() : ""; }
extern "C" int hg_multi_peer_access(const hg_multi *m, int from_index, int to_index)
{
    const int G = m ? (int)m->devs.size() : 0;
    if (from_index < 0 || to_index < 0 || from_index >= G || to_index >= G) return -1;
    return m->access[(size_t)from_index * G + to_index];
}

// setImage (:290-316) for every device: the reference's single source image, shared by all frames of a batch.
extern "C" int hg_multi_set_image(hg_multi *m, const uint8_t *rgba, int w, int h)
{
    if (!m) return mfail(nullptr, HG_ERR_INVALID, "multi is NULL");
    if (!rgba || w <= 0 || h <= 0) return mfail(m, HG_ERR_INVALID, "hg_multi_set_image: bad image");
    const size_t bytes = (size_t)w * h * 4;
    const int G = (int)m->devs.size();
    // Queued warps still read the old image, and the previous fan-out may still be running on the peers' copy streams (set_image
    // returns after the root's H2D): EVERY device is settled -- its warp stream waited for its own "whole image here" event, so
    // a settled device has finished its pulls -- before ANY buffer is replaced: peers pull slices out of each other's buffers.
    for (auto &d : m->devs) {
        MHG(m, d.ctx, hg_sync(d.ctx));
        MHIP(m, hipSetDevice(d.id));
        MHIP(m, hipStreamSynchronize(d.copy));
    }
    for (auto &d : m->devs) {
        if (bytes > d.img_cap) {
            // grow: new buffer first, the context's alias moves to it, only then the old one goes (never a dangling alias,
            // whatever fails on the way)
            MHIP(m, hipSetDevice(d.id));
            void *q = nullptr;
            if (hipMalloc(&q, bytes) != hipSuccess) return mfail(m, HG_ERR_NOMEM, "hg_multi_set_image: hipMalloc failed");
            const int rc = hg_set_image_device(d.ctx, q, w, h);
            if (rc != HG_OK) { (void)hipFree(q); return mfail(m, rc, hg_last_error(d.ctx)); }
            if (d.d_img) MHIP(m, hipFree(d.d_img));
            d.d_img = static_cast<uint8_t *>(q); d.img_cap = bytes;
        }
    }
    auto &root = m->devs[0];
    MHIP(m, hipSetDevice(root.id));
    {
        MHIP(m, hipMemcpyAsync(root.d_img, rgba, bytes, hipMemcpyHostToDevice, root.copy));
        MHIP(m, hipEventRecord(root.have_slice, root.copy));
        MHIP(m, hipEventRecord(root.have_image, root.copy));
        // slices: 4-byte aligned, the last one takes the remainder
        const size_t slice = ((bytes / G) + 3) & ~(size_t)3;
        auto lo = [&](int k) { return std::min(bytes, slice * (size_t)k); };
        auto hi = [&](int k) { return k == G - 1 ? bytes : std::min(bytes, slice * (size_t)(k + 1)); };
        std::vector<FanOp> plan;
        plan_fanout(G, m->access.data(), plan);
        std::vector<char> reads_host((size_t)G, 0);
        int phase = 0;
        auto close_scatter = [&]() -> int {                  // every peer's "own slice here" event, once its scatter copy is queued
            for (int p = 1; p < G; p++) { auto &d = m->devs[p]; MHIP(m, hipSetDevice(d.id)); MHIP(m, hipEventRecord(d.have_slice, d.copy)); }
            return HG_OK;
        };
        for (const FanOp &op : plan) {
            if (op.phase == 1 && phase == 0) { phase = 1; const int rc = close_scatter(); if (rc != HG_OK) return rc; }
            auto &dq = m->devs[op.dst];
            if (hi(op.slice) <= lo(op.slice)) continue;
            MHIP(m, hipSetDevice(dq.id));
            if (op.src < 0) {                                // no route between the devices: this slice comes from the caller's buffer
                MHIP(m, hipMemcpyAsync(dq.d_img + lo(op.slice), rgba + lo(op.slice), hi(op.slice) - lo(op.slice), hipMemcpyHostToDevice, dq.copy));
                reads_host[(size_t)op.dst] = 1;
            } else {
                auto &dp = m->devs[op.src];
                MHIP(m, hipStreamWaitEvent(dq.copy, dp.have_slice, 0));      // (the root's have_slice covers the whole image)
                MHIP(m, hipMemcpyPeerAsync(dq.d_img + lo(op.slice), dq.id, dp.d_img + lo(op.slice), dp.id, hi(op.slice) - lo(op.slice), dq.copy));
            }
        }
        if (phase == 0) { const int rc = close_scatter(); if (rc != HG_OK) return rc; }
        for (int q = 1; q < G; q++) { auto &dq = m->devs[q]; MHIP(m, hipSetDevice(dq.id)); MHIP(m, hipEventRecord(dq.have_image, dq.copy)); }
        // The host waits for the copies that read ITS buffer only (caller memory is not retained): the root's H2D, and the streams of
        // devices that had to take slices from the host.  The fan-out over xGMI keeps running: every device's warp stream waits for
        // ITS have_image event, so the root's first frames overlap the scatter + all-gather.
        MHIP(m, hipSetDevice(root.id));
        MHIP(m, hipEventSynchronize(root.have_image));
        for (int q = 1; q < G; q++) if (reads_host[(size_t)q]) { MHIP(m, hipSetDevice(m->devs[q].id)); MHIP(m, hipStreamSynchronize(m->devs[q].copy)); }
    }
    for (auto &d : m->devs) {
        MHG(m, d.ctx, hg_set_image_device(d.ctx, d.d_img, w, h));
        MHG(m, d.ctx, hg_stream_wait_event(d.ctx, d.have_image));
    }
    m->W = w; m->H = h;
    return HG_OK;
}

extern "C" int hg_multi_piecewise_set_mesh(hg_multi *m, const float *src_points, int n_points, const uint32_t *triangles, int n_triangles,
                                           int min_src_x, int min_src_y)
{
    if (!m) return mfail(nullptr, HG_ERR_INVALID, "multi is NULL");
    for (auto &d : m->devs) MHG(m, d.ctx, hg_piecewise_set_mesh(d.ctx, src_points, n_points, triangles, n_triangles, min_src_x, min_src_y));
    m->n_pts = n_points;
    return HG_OK;
}

// The caller loop for F frames over all devices: device i warps the contiguous block hg_multi_partition gives it, all
// devices at once.  Frames stay resident (hg_multi_frame) unless out_host is given: then out_host[f] receives frame f
// (4*obj_w*obj_h bytes; pinned memory from hg_host_alloc makes the copies of different devices overlap).
static void reset_partition(hg_multi *m)
{
    for (auto &d : m->devs) { d.first = 0; d.count = 0; d.offs.assign(1, 0); }
    m->geoms.clear();
}

// one source per frame ("replicas only", SURVEY.md §8e / README.md:121-137): device i receives the images of ITS block only
static int upload_block_images(hg_multi *m, hg_multi::Dev &d, const uint8_t *const *images, int w, int h)
{
    const size_t bytes = (size_t)w * h * 4;
    MHG(m, d.ctx, hg_sync(d.ctx));                           // queued warps still read the previous sources
    if (bytes * (size_t)d.count > d.imgs_cap) {
        MHIP(m, hipSetDevice(d.id));
        void *q = nullptr;
        if (hipMalloc(&q, bytes * (size_t)d.count) != hipSuccess) return mfail(m, HG_ERR_NOMEM, "hg_multi: hipMalloc of the per-frame sources failed");
        const int rc = hg_set_images_device(d.ctx, q, w, h, d.count, bytes);     // the context's alias moves first
        if (rc != HG_OK) { (void)hipFree(q); return mfail(m, rc, hg_last_error(d.ctx)); }
        if (d.d_imgs) MHIP(m, hipFree(d.d_imgs));
        d.d_imgs = static_cast<uint8_t *>(q); d.imgs_cap = bytes * (size_t)d.count;
    }
    MHIP(m, hipSetDevice(d.id));
    for (int k = 0; k < d.count; k++) {
        if (!images[d.first + k]) return mfail(m, HG_ERR_INVALID, "hg_multi: images[f] is NULL");
        MHIP(m, hipMemcpyAsync(d.d_imgs + bytes * (size_t)k, images[d.first + k], bytes, hipMemcpyHostToDevice, d.copy));
    }
    MHIP(m, hipEventRecord(d.have_image, d.copy));
    return HG_OK;
}

// frames leave the devices: the copies of EVERY device are queued (each behind its own kernels) before any device is waited
// for; a device whose settlement then redid frames through the map path gets those frames copied again.
static int drain_to_host(hg_multi *m, const hg_geom *geoms, uint8_t *const *out_host)
{
    std::vector<long> redone0(m->devs.size(), 0);
    if (out_host) {
        for (size_t i = 0; i < m->devs.size(); i++) {
            auto &d = m->devs[i];
            redone0[i] = hg_redone_frames(d.ctx);
            for (int k = 0; k < d.count; k++) {
                const hg_geom &g = geoms[d.first + k];
                const size_t bytes = (g.obj_w > 0 && g.obj_h > 0) ? (size_t)g.obj_w * g.obj_h * 4 : 0;
                if (!bytes) continue;
                if (!out_host[d.first + k]) return mfail(m, HG_ERR_INVALID, "hg_multi: out_host[f] is NULL");
                MHG(m, d.ctx, hg_enqueue_copy_to_host(d.ctx, out_host[d.first + k], d.d_out + d.offs[k], bytes));
            }
        }
    }
    for (size_t i = 0; i < m->devs.size(); i++) {
        auto &d = m->devs[i];
        if (!d.count) continue;
        MHG(m, d.ctx, hg_sync(d.ctx));
        if (out_host && hg_redone_frames(d.ctx) != redone0[i]) {         // rare: flagged frames were rewritten after their copy
            for (int k = 0; k < d.count; k++) {
                const hg_geom &g = geoms[d.first + k];
                const size_t bytes = (g.obj_w > 0 && g.obj_h > 0) ? (size_t)g.obj_w * g.obj_h * 4 : 0;
                if (bytes) MHG(m, d.ctx, hg_enqueue_copy_to_host(d.ctx, out_host[d.first + k], d.d_out + d.offs[k], bytes));
            }
            MHG(m, d.ctx, hg_sync(d.ctx));
        }
    }
    return HG_OK;
}

// The caller loop for F frames over all devices: device i warps the contiguous block hg_multi_partition gives it, all
// devices at once.  Frames stay resident (hg_multi_frame) unless out_host is given: then out_host[f] receives frame f
// (4*obj_w*obj_h bytes; pinned memory from hg_host_alloc makes the copies of different devices overlap).
// images != NULL: one source per frame, images[f] = width x height RGBA8 on the host; every device uploads its own block.
static int multi_batch_impl(hg_multi *m, const float *dst_points, const hg_geom *geoms, int n_frames, const uint8_t *const *images, int w, int h,
                            uint8_t *const *out_host)
{
    if (!m) return mfail(nullptr, HG_ERR_INVALID, "multi is NULL");
    reset_partition(m);                                      // (a failure below must not leave the previous batch's blocks behind)
    if (!dst_points || !geoms || n_frames <= 0) return mfail(m, HG_ERR_INVALID, "hg_multi_warp_piecewise_batch: bad arguments");
    if (m->n_pts <= 0) return mfail(m, HG_ERR_STATE, "no mesh: call hg_multi_piecewise_set_mesh first");
    if (images ? (w <= 0 || h <= 0) : m->W <= 0) return mfail(m, HG_ERR_STATE, images ? "bad per-frame image size" : "no source image: call hg_multi_set_image first");
    const int G = (int)m->devs.size();
    m->geoms.assign(geoms, geoms + n_frames);
    // 1. enqueue every device's share (asynchronous: all devices compute at the same time)
    for (int i = 0; i < G; i++) {
        auto &d = m->devs[i];
        hg_multi_partition(n_frames, G, i, &d.first, &d.count);
        d.offs.assign((size_t)std::max(d.count, 1), 0);
        if (d.count == 0) continue;
        size_t total = 0;
        MHG(m, nullptr, hg_pack_offsets(geoms + d.first, d.count, d.offs.data(), &total));
        MHG(m, d.ctx, ensure_dev(m, d, d.d_out, d.out_cap, std::max<size_t>(total, 256)));
        if (images) {
            MHG(m, d.ctx, upload_block_images(m, d, images, w, h));
            MHG(m, d.ctx, hg_set_images_device(d.ctx, d.d_imgs, w, h, d.count, (size_t)w * h * 4));
            MHG(m, d.ctx, hg_stream_wait_event(d.ctx, d.have_image));
        }
        MHG(m, d.ctx, hg_warp_inverse_piecewise_batch_device(d.ctx, dst_points + (size_t)d.first * m->n_pts * 2, geoms + d.first, d.offs.data(), d.count, d.d_out));
    }
    if (images)                                              // caller memory is not retained: the H2D copies have left it
        for (auto &d : m->devs) if (d.count) { MHIP(m, hipSetDevice(d.id)); MHIP(m, hipStreamSynchronize(d.copy)); }
    // 2. + 3. frames leave the devices; settle (also redoes frames the fused path flagged)
    return drain_to_host(m, geoms, out_host);
}

static int multi_fail_settle(hg_multi *m, int rc)
{
    if (rc != HG_OK && m) {                                  // nothing may still be writing into the caller's buffers when an error is reported
        const std::string why = m->err;
        for (auto &d : m->devs) if (d.ctx) (void)hg_sync(d.ctx);
        reset_partition(m);
        m->err = why; g_merr = why;
    }
    return rc;
}

extern "C" int hg_multi_warp_piecewise_batch(hg_multi *m, const float *dst_points, const hg_geom *geoms, int n_frames, uint8_t *const *out_host)
{
    return multi_fail_settle(m, multi_batch_impl(m, dst_points, geoms, n_frames, nullptr, 0, 0, out_host));
}

extern "C" int hg_multi_warp_piecewise_batch_images(hg_multi *m, const float *dst_points, const hg_geom *geoms, int n_frames,
                                                    const uint8_t *const *images, int width, int height, uint8_t *const *out_host)
{
    if (!images) return mfail(m, HG_ERR_INVALID, "hg_multi_warp_piecewise_batch_images: images is NULL");
    const int rc = multi_fail_settle(m, multi_batch_impl(m, dst_points, geoms, n_frames, images, width, height, out_host));
    if (m) { m->W = 0; m->H = 0; }                           // the shared image (hg_multi_set_image) is no longer what the contexts read
    return rc;
}

// The same for affine / projective frames given as point sets (hg_geometric_set_frames_points: the per-frame solves run on each
// device for its own block of frames).
static int multi_geo_impl(hg_multi *m, int kind, const float *from, const float *to, const hg_geom *geoms, int n_frames,
                          const uint8_t *const *images, int w, int h, uint8_t *const *out_host)
{
    if (!m) return mfail(nullptr, HG_ERR_INVALID, "multi is NULL");
    reset_partition(m);
    if ((kind != HG_AFFINE && kind != HG_PROJECTIVE) || !from || !to || !geoms || n_frames <= 0) return mfail(m, HG_ERR_INVALID, "hg_multi_warp_geometric_batch: bad arguments");
    if (images ? (w <= 0 || h <= 0) : m->W <= 0) return mfail(m, HG_ERR_STATE, images ? "bad per-frame image size" : "no source image: call hg_multi_set_image first");
    const int G = (int)m->devs.size();
    const size_t per = kind == HG_AFFINE ? 6 : 8;
    m->geoms.assign(geoms, geoms + n_frames);
    for (int i = 0; i < G; i++) {
        auto &d = m->devs[i];
        hg_multi_partition(n_frames, G, i, &d.first, &d.count);
        d.offs.assign((size_t)std::max(d.count, 1), 0);
        if (d.count == 0) continue;
        size_t total = 0;
        MHG(m, nullptr, hg_pack_offsets(geoms + d.first, d.count, d.offs.data(), &total));
        MHG(m, d.ctx, ensure_dev(m, d, d.d_out, d.out_cap, std::max<size_t>(total, 256)));
        if (images) {
            MHG(m, d.ctx, upload_block_images(m, d, images, w, h));
            MHG(m, d.ctx, hg_set_images_device(d.ctx, d.d_imgs, w, h, d.count, (size_t)w * h * 4));
            MHG(m, d.ctx, hg_stream_wait_event(d.ctx, d.have_image));
        }
        MHG(m, d.ctx, hg_geometric_set_frames_points(d.ctx, kind, from + (size_t)d.first * per, to + (size_t)d.first * per, geoms + d.first, d.offs.data(), d.count));
        MHG(m, d.ctx, hg_warp_inverse_geometric_frames_device(d.ctx, d.d_out));
    }
    if (images)
        for (auto &d : m->devs) if (d.count) { MHIP(m, hipSetDevice(d.id)); MHIP(m, hipStreamSynchronize(d.copy)); }
    return drain_to_host(m, geoms, out_host);
}

extern "C" int hg_multi_warp_geometric_batch(hg_multi *m, int kind, const float *from, const float *to, const hg_geom *geoms, int n_frames, uint8_t *const *out_host)
{
    return multi_fail_settle(m, multi_geo_impl(m, kind, from, to, geoms, n_frames, nullptr, 0, 0, out_host));
}

extern "C" int hg_multi_warp_geometric_batch_images(hg_multi *m, int kind, const float *from, const float *to, const hg_geom *geoms, int n_frames,
                                                    const uint8_t *const *images, int width, int height, uint8_t *const *out_host)
{
    if (!images) return mfail(m, HG_ERR_INVALID, "hg_multi_warp_geometric_batch_images: images is NULL");
    const int rc = multi_fail_settle(m, multi_geo_impl(m, kind, from, to, geoms, n_frames, images, width, height, out_host));
    if (m) { m->W = 0; m->H = 0; }
    return rc;
}

extern "C" int hg_multi_frame(hg_multi *m, int frame, int *device_index, void **d_ptr, size_t *bytes)
{
    if (!m) return mfail(nullptr, HG_ERR_INVALID, "multi is NULL");
    if (frame < 0 || (size_t)frame >= m->geoms.size()) return mfail(m, HG_ERR_INVALID, "hg_multi_frame: no such frame in the last batch");
    for (size_t i = 0; i < m->devs.size(); i++) {
        const auto &d = m->devs[i];
        if (frame >= d.first && frame < d.first + d.count) {
            const hg_geom &g = m->geoms[frame];
            if (device_index) *device_index = (int)i;
            if (d_ptr) *d_ptr = d.d_out + d.offs[frame - d.first];
            if (bytes) *bytes = (g.obj_w > 0 && g.obj_h > 0) ? (size_t)g.obj_w * g.obj_h * 4 : 0;
            return HG_OK;
        }
    }
    return mfail(m, HG_ERR_INVALID, "hg_multi_frame: no such frame in the last batch");
}
