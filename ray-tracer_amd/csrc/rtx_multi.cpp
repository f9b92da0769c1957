// rtx_multi.cpp -- one frame on several GPUs of a node from a single process, behind the C ABI
// (include/rtx.h, rtx_multi_*).  Stands in for the Rayon loop over 10-row bands of
// Camera::parallel_render (src/scenes/camera.rs:177-193): the World is copied to every GPU once,
// band b of the image is rendered by GPU b mod n (every GPU gets its share of cheap and of dear
// rows), nothing travels between GPUs while they render, and each GPU then sends its bands to
// their places in the frame on the first GPU with one strided device-to-device copy over xGMI.
// Built on the single-GPU entry points only (rtx_context_create, rtx_scene_upload,
// rtx_render_part_device); host code, no arithmetic of the render loop.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rtx.h"

namespace {

struct Device {
    int id = 0;
    rtx_context *ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t start = nullptr, rendered = nullptr, done = nullptr;
    void *part = nullptr;  // this GPU's bands, back to back
    size_t part_bytes = 0;
};

int report(int code, const std::string &what);

}  // namespace

struct rtx_multi {
    std::vector<Device> dev;
    void *frame = nullptr;  // the assembled frame on dev[0] when the caller wants it in host memory
    size_t frame_bytes = 0;
    uint32_t band_rows = 10;  // src/scenes/camera.rs:177
};

struct rtx_multi_scene {
    rtx_multi *owner = nullptr;
    std::vector<rtx_scene *> scene;  // one upload per GPU
};

namespace {

thread_local std::string tl_multi_error;

int report(int code, const std::string &what)
{
    tl_multi_error = what;
    return code;
}
int report_hip(hipError_t e, const char *what)
{
    return report(RTX_ERR_DEVICE, std::string(what) + " failed: " + hipGetErrorString(e));
}
// a failed single-GPU call left its message with rtx_last_error()
int forward(int rc, int device)
{
    return report(rc, "GPU " + std::to_string(device) + ": " + rtx_last_error());
}

#define MULTI_HIP(expr)                                   \
    do {                                                  \
        hipError_t e_ = (expr);                           \
        if (e_ != hipSuccess) return report_hip(e_, #expr); \
    } while (0)

int grow(void **p, size_t *have, size_t need)
{
    if (need <= *have) return RTX_OK;
    if (*p) MULTI_HIP(hipFree(*p));
    *p = nullptr;
    *have = 0;
    MULTI_HIP(hipMalloc(p, need));
    *have = need;
    return RTX_OK;
}

}  // namespace

extern "C" {

const char *rtx_multi_last_error(void) { return tl_multi_error.c_str(); }

int rtx_multi_create(const int *device_ids, int n_devices, rtx_multi **out)
{
    if (!out) return report(RTX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return report(RTX_ERR_INVALID, "between 1 and 64 devices are needed");
    int prev = 0;
    (void)hipGetDevice(&prev);
    std::unique_ptr<rtx_multi> m(new rtx_multi);
    m->dev.resize((size_t)n_devices);
    auto build = [&]() -> int {
        for (int k = 0; k < n_devices; ++k) {
            Device &d = m->dev[(size_t)k];
            d.id = device_ids[k];
            int rc = rtx_context_create(d.id, &d.ctx);
            if (rc) return forward(rc, d.id);
            MULTI_HIP(hipSetDevice(d.id));
            MULTI_HIP(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
            MULTI_HIP(hipEventCreate(&d.start));
            MULTI_HIP(hipEventCreate(&d.rendered));
            MULTI_HIP(hipEventCreate(&d.done));
        }
        // every GPU writes its bands into the first GPU's memory
        for (int k = 1; k < n_devices; ++k) {
            const int a = m->dev[(size_t)k].id, root = m->dev[0].id;
            if (a == root) continue;
            int can = 0;
            MULTI_HIP(hipDeviceCanAccessPeer(&can, a, root));
            if (!can) continue;  // the copies then go through host memory: slower, still right
            MULTI_HIP(hipSetDevice(a));
            const hipError_t e = hipDeviceEnablePeerAccess(root, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return report_hip(e, "hipDeviceEnablePeerAccess");
            (void)hipGetLastError();
        }
        return RTX_OK;
    };
    const int rc = build();
    (void)hipSetDevice(prev);
    if (rc) {
        rtx_multi_destroy(m.release());
        return rc;
    }
    *out = m.release();
    return RTX_OK;
}

void rtx_multi_destroy(rtx_multi *m)
{
    if (!m) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (size_t k = 0; k < m->dev.size(); ++k) {
        Device &d = m->dev[k];
        (void)hipSetDevice(d.id);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        if (d.part) (void)hipFree(d.part);
        if (k == 0 && m->frame) (void)hipFree(m->frame);
        for (hipEvent_t e : { d.start, d.rendered, d.done })
            if (e) (void)hipEventDestroy(e);
        if (d.stream) (void)hipStreamDestroy(d.stream);
        if (d.ctx) rtx_context_destroy(d.ctx);
    }
    (void)hipSetDevice(prev);
    delete m;
}

int rtx_multi_device_count(const rtx_multi *m) { return m ? (int)m->dev.size() : 0; }

int rtx_multi_scene_upload(rtx_multi *m, const rtx_scene_desc *desc, rtx_multi_scene **out)
{
    if (!m || !desc || !out) return report(RTX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    std::unique_ptr<rtx_multi_scene> s(new rtx_multi_scene);
    s->owner = m;
    s->scene.assign(m->dev.size(), nullptr);
    for (size_t k = 0; k < m->dev.size(); ++k) {
        const int rc = rtx_scene_upload(m->dev[k].ctx, desc, &s->scene[k]);
        if (rc) {
            const int code = forward(rc, m->dev[k].id);
            rtx_multi_scene_destroy(s.release());
            return code;
        }
    }
    *out = s.release();
    return RTX_OK;
}

void rtx_multi_scene_destroy(rtx_multi_scene *s)
{
    if (!s) return;
    for (rtx_scene *one : s->scene)
        if (one) rtx_scene_destroy(one);
    delete s;
}

int rtx_multi_render(rtx_multi *m, const rtx_multi_scene *scene, const rtx_camera_desc *camera, uint32_t max_depth, int precision,
                     uint32_t format, void *out, size_t out_bytes, int out_on_first_device, uint32_t flags, rtx_multi_stats *stats)
{
    if (!m || !scene || scene->owner != m) return report(RTX_ERR_INVALID, "scene does not belong to this set of devices");
    if (!camera || camera->hsize == 0 || camera->vsize == 0) return report(RTX_ERR_INVALID, "camera is NULL or has an empty canvas");
    if (precision != RTX_F64 && precision != RTX_F32) return report(RTX_ERR_INVALID, "unknown precision");
    const size_t elem = precision == RTX_F64 ? sizeof(double) : sizeof(float);
    const size_t pixel_bytes = format ? (size_t)(format & 0xffu) : 3 * elem;
    const size_t row_bytes = (size_t)camera->hsize * pixel_bytes;
    const size_t need = row_bytes * camera->vsize;
    if (!out || out_bytes < need) return report(RTX_ERR_INVALID, "output buffer too small for the frame");
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t n = (uint32_t)m->dev.size(), band = m->band_rows, H = camera->vsize;
    int prev = 0;
    (void)hipGetDevice(&prev);
    struct Restore {
        int dev;
        ~Restore() { (void)hipSetDevice(dev); }
    } restore{ prev };

    char *frame = (char *)out;
    if (!out_on_first_device) {
        MULTI_HIP(hipSetDevice(m->dev[0].id));
        int rc = grow(&m->frame, &m->frame_bytes, need);
        if (rc) return rc;
        frame = (char *)m->frame;
    }
    if (stats) std::memset(stats, 0, sizeof *stats);
    std::vector<rtx_stats> counted(n);
    const bool count = stats && (flags & RTX_MULTI_COUNT_RAYS);
    // every GPU: its bands, then one strided copy that drops them into the frame
    auto enqueue = [&](uint32_t k) -> int {
        Device &d = m->dev[k];
        const uint32_t rows = rtx_part_rows(H, band, n, k);
        if (rows == 0) return RTX_OK;
        MULTI_HIP(hipSetDevice(d.id));
        int rc = grow(&d.part, &d.part_bytes, (size_t)rows * row_bytes);
        if (rc) return rc;
        MULTI_HIP(hipEventRecord(d.start, d.stream));
        rc = rtx_render_part_device(d.ctx, scene->scene[k], camera, max_depth, precision, band, n, k, format, d.part,
                                    (size_t)rows * row_bytes, d.stream, count ? &counted[k] : nullptr);
        if (rc) return forward(rc, d.id);
        MULTI_HIP(hipEventRecord(d.rendered, d.stream));
        const size_t band_bytes = (size_t)band * row_bytes;
        const uint32_t whole = rows / band, rest = rows % band;  // only the image's last band can be short
        char *dst = frame + (size_t)k * band_bytes;
        if (whole)
            MULTI_HIP(hipMemcpy2DAsync(dst, (size_t)n * band_bytes, d.part, band_bytes, band_bytes, whole, hipMemcpyDeviceToDevice, d.stream));
        if (rest)
            MULTI_HIP(hipMemcpyAsync(dst + (size_t)whole * n * band_bytes, (char *)d.part + (size_t)whole * band_bytes,
                                     (size_t)rest * row_bytes, hipMemcpyDeviceToDevice, d.stream));
        MULTI_HIP(hipEventRecord(d.done, d.stream));
        return RTX_OK;
    };
    // A failure on one GPU must not return while the others still render and copy into `out`: the caller may
    // free it.  Everything started so far (the failing GPU included) is waited for first; the first error is kept.
    auto drain = [&](uint32_t started) {
        const std::string first = tl_multi_error;
        for (uint32_t j = 0; j < started; ++j) {
            Device &d = m->dev[j];
            if (hipSetDevice(d.id) != hipSuccess) continue;
            (void)hipStreamSynchronize(d.stream);
            (void)rtx_context_synchronize(d.ctx);
        }
        (void)hipGetLastError();
        tl_multi_error = first;
    };
    for (uint32_t k = 0; k < n; ++k) {
        const int rc = enqueue(k);
        if (rc) {
            drain(k + 1);
            return rc;
        }
    }
    double render_ms = 0.0, gather_ms = 0.0;
    auto wait_for = [&](uint32_t k) -> int {
        Device &d = m->dev[k];
        if (rtx_part_rows(H, band, n, k) == 0) return RTX_OK;
        MULTI_HIP(hipSetDevice(d.id));
        MULTI_HIP(hipEventSynchronize(d.done));
        if (stats) {
            float a = 0.f, b = 0.f;
            MULTI_HIP(hipEventElapsedTime(&a, d.start, d.rendered));
            MULTI_HIP(hipEventElapsedTime(&b, d.rendered, d.done));
            render_ms = std::max(render_ms, (double)a);
            gather_ms = std::max(gather_ms, (double)b);
            stats->color_rays += counted[k].color_rays;
            stats->shadow_rays += counted[k].shadow_rays;
        }
        return RTX_OK;
    };
    for (uint32_t k = 0; k < n; ++k) {
        const int rc = wait_for(k);
        if (rc) {
            drain(n);
            return rc;
        }
    }
    int overflow = RTX_OK;
    for (uint32_t k = 0; k < n; ++k) {  // an overflow of a ray queue on any GPU spoils the frame (every GPU is asked: the report clears it)
        const int rc = rtx_context_synchronize(m->dev[k].ctx);
        if (rc && !overflow) overflow = forward(rc, m->dev[k].id);
    }
    if (overflow) return overflow;
    if (!out_on_first_device) {
        MULTI_HIP(hipSetDevice(m->dev[0].id));
        MULTI_HIP(hipMemcpy(out, frame, need, hipMemcpyDeviceToHost));
    }
    if (stats) {
        stats->n_devices = n;
        stats->render_ms = render_ms;
        stats->gather_ms = gather_ms;
        stats->gather_bytes = need - (size_t)rtx_part_rows(H, band, n, 0) * row_bytes;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RTX_OK;
}

}  // extern "C"
