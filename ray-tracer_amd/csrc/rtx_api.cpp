// rtx_api.cpp -- the C ABI declared in include/rtx.h: contexts, scene upload (repacking the
// caller's structure-of-arrays World into the device object table) and the render calls,
// which drive the kernels of rt_kernels.h level by level.  Host code only; no arithmetic of
// the render loop happens here apart from the camera origin, which the reference recomputes
// per pixel (src/scenes/camera.rs:154) and which is the same value for every pixel.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/rtx.h"
#include "rt_launch.h"

using namespace rtx;

namespace {

thread_local std::string tl_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    tl_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(RTX_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr uint32_t kMaxLevels = 8;  // rtx_stats.rays_per_level entries; max_depth <= 7

// Device memory the kernels of one frame use beyond the scene tables and the framebuffer.  A
// slot's workspace grows on demand (hipMalloc); the allocation-free entry point carves a fixed one
// out of the file-scope device array of scratch.hip.
struct Workspace {
    void *queues = nullptr;     // both ping-pong ray queues (measured frames: the queue of the odd levels)
    size_t queue_bytes = 0;
    void *queues_even = nullptr;  // measured frames only: the queue of the even levels
    size_t queue_even_bytes = 0;
    uint32_t *host_counts = nullptr;  // pinned and mapped: a sequence number, then the sub-queue lengths of one level (measured frames)
    uint32_t *host_counts_dev = nullptr;  // the same memory as the device addresses it
    uint32_t publish_seq = 0;
    uint32_t *counts = nullptr; // sub-queue length counters: per chunk, per level, per sub-queue
    size_t counts_n = 0;        // in words
    void *prim = nullptr;       // primary-ray tables of the scene and camera origin below
    size_t prim_bytes = 0;
    StatShard *shards = nullptr;
    bool fixed = false;         // not ours to grow or free
    uint64_t prim_scene = 0;    // rtx_scene::serial the tables were built from (0: none)
    double prim_origin[3] = { 0, 0, 0 };
    size_t prim_elem = 0;       // sizeof(R) of the tables
    uint32_t n_sub = 1;         // sub-queues per queue of the last frame enqueued
};

// Scratch of one in-flight render call.  A context keeps a pool of these so that calls from
// different host threads overlap on the device instead of serialising.
struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
    Workspace ws;
    void *fb = nullptr;         // device framebuffer for host-output calls
    size_t fb_bytes = 0;
    void *aux = nullptr;        // ray upload / probe output / 8-bit pixels
    size_t aux_bytes = 0;
    hipEvent_t done = nullptr;  // recorded when a call returns with work still in flight
    bool pending = false;
    bool busy = false;
    hipStream_t active = nullptr;  // stream this call may have queued work on that nothing has waited for yet
    bool has_active = false;
    char *pin[2] = { nullptr, nullptr };  // pinned staging buffers for large copies to pageable memory
    hipEvent_t pin_ev[2] = { nullptr, nullptr };

    // Give everything back to the runtime (the slot's device must be current).
    void release()
    {
        if (stream) (void)hipStreamSynchronize(stream);
        for (void *p : { ws.queues, ws.queues_even, fb, aux, ws.prim, (void *)ws.counts, (void *)ws.shards })
            if (p) (void)hipFree(p);
        if (ws.host_counts) (void)hipHostFree(ws.host_counts);
        ws = Workspace();
        fb = aux = nullptr;
        if (done) (void)hipEventDestroy(done);
        done = nullptr;
        for (int k = 0; k < 2; ++k) {
            if (pin[k]) (void)hipHostFree(pin[k]);
            if (pin_ev[k]) (void)hipEventDestroy(pin_ev[k]);
            pin[k] = nullptr;
            pin_ev[k] = nullptr;
        }
        for (auto &e : ev) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
};

}  // namespace

struct rtx_context {
    int device = 0;
    int cu_count = 256;
    int clock_mhz = 0;
    std::string name;
    size_t workspace_limit = (size_t)8 << 30;
    uint32_t cull_min_objects = 16;  // scenes with at least this many bounded objects get the bounding-sphere tests
    uint32_t blocks_per_cu = 8;
    uint32_t bounce_blocks = 7680;   // workgroups of one bounce launch (set from the device's size at creation)
    uint32_t bounce_blocks_cull = 65536u;  // the same for scenes with bounding-sphere tests: one per piece the queue can hold, up to this many
    uint32_t bounce_blocks_lane = 512u;    // and for the kernel that keeps the packed hierarchy in LDS (one resident workgroup per CU)
    bool lane_tree = true;                 // scenes uploaded from now on may use that kernel
    void *scratch = nullptr;         // this device's copy of the file-scope array of scratch.hip
    size_t scratch_bytes = 0;
    uint32_t *overflow_host = nullptr;  // pinned word the kernels set when a ray queue overflows
    uint32_t *overflow_dev = nullptr;   // the same word as the device addresses it
    uint32_t debug_queue_divisor = 1;   // tests: queues this many times smaller than planned
    size_t measured_above = (size_t)8 << 30;  // frames whose worst-case queues exceed this are sized level by level
    std::mutex mu;
    std::vector<std::unique_ptr<Slot>> slots;
};

struct rtx_scene {
    rtx_context *ctx = nullptr;
    uint64_t serial = 0;  // unique per upload: identifies the tables that derived data was built from
    uint32_t n = 0;
    uint32_t sf = 0;
    void *blob64 = nullptr, *blob32 = nullptr;  // device tables, one allocation per precision
    SceneDev<double> dev64;
    SceneDev<float> dev32;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int slot_acquire(rtx_context *ctx, Slot **out)
{
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (auto &s : ctx->slots)
        if (!s->busy) {
            s->busy = true;
            *out = s.get();
            return RTX_OK;
        }
    std::unique_ptr<Slot> s(new Slot);
    auto build = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        for (auto &e : s->ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
        HIP_TRY(hipMalloc((void **)&s->ws.shards, sizeof(StatShard) * kStatShards));
        return RTX_OK;
    };
    const int rc = build();
    if (rc) {
        s->release();  // what a half-built slot holds
        return rc;
    }
    s->busy = true;
    *out = s.get();
    ctx->slots.push_back(std::move(s));
    return RTX_OK;
}
void slot_release(rtx_context *ctx, Slot *s)
{
    std::lock_guard<std::mutex> lk(ctx->mu);
    s->busy = false;
}
struct SlotLease {
    rtx_context *ctx;
    Slot *slot = nullptr;
    explicit SlotLease(rtx_context *c) : ctx(c) {}
    ~SlotLease()
    {
        if (!slot) return;
        // A call that fails half-way may leave kernels running on the slot's scratch memory: the
        // next user of the slot must wait for them like for any work left in flight.
        if (slot->has_active && hipEventRecord(slot->done, slot->active) == hipSuccess) slot->pending = true;
        slot->has_active = false;
        slot_release(ctx, slot);
    }
};

// An earlier call may have returned with work still in flight that uses this slot's scratch
// memory, possibly on another stream: whatever `stream` runs next waits for it on the device.
int slot_order_after_pending(Slot *slot, hipStream_t stream)
{
    if (slot->pending) {
        HIP_TRY(hipStreamWaitEvent(stream, slot->done, 0));
        slot->pending = false;
    }
    slot->active = stream;
    slot->has_active = true;
    return RTX_OK;
}
// Everything the call queued has been waited for (or handed over through slot->done).
void slot_settled(Slot *slot) { slot->has_active = false; }

int grow(void **ptr, size_t *have, size_t need)
{
    if (need <= *have) return RTX_OK;
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(ptr, need));
    *have = need;
    return RTX_OK;
}

// The ray queues of slots that no call is using and nothing in flight refers to: given back when
// another call cannot get its own (queues are sized for the worst case and kept for reuse, so a
// few large frames on several host threads can hold tens of GB between them).
void trim_idle_queues(rtx_context *ctx, const Workspace *keep)
{
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (auto &s : ctx->slots) {
        if (s->busy || &s->ws == keep) continue;
        // work a call left in flight may have finished since: then nothing refers to the queues any more
        if (s->pending && hipEventQuery(s->done) == hipSuccess) s->pending = false;
        if (s->pending) continue;
        if (s->ws.queues) (void)hipFree(s->ws.queues);
        if (s->ws.queues_even) (void)hipFree(s->ws.queues_even);
        s->ws.queues = s->ws.queues_even = nullptr;
        s->ws.queue_bytes = s->ws.queue_even_bytes = 0;
    }
}

// Device -> pageable host memory.  hipMemcpy stages such a copy through pinned memory with one
// CPU thread (about 10-18 GB/s here); two pinned buffers, the DMA of one overlapping the memcpy
// out of the other, and four threads on the memcpy reach about 50 GB/s.  Blocks until the data
// is in dst.
constexpr size_t kPinBytes = (size_t)16 << 20;
constexpr int kCopyThreads = 4;
int copy_to_host(Slot *slot, hipStream_t stream, void *dst, const void *src, size_t bytes)
{
    if (bytes < 2 * kPinBytes) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return RTX_OK;
    }
    for (int k = 0; k < 2; ++k) {
        if (!slot->pin[k]) HIP_TRY(hipHostMalloc((void **)&slot->pin[k], kPinBytes, hipHostMallocDefault));
        if (!slot->pin_ev[k]) HIP_TRY(hipEventCreateWithFlags(&slot->pin_ev[k], hipEventDisableTiming));
    }
    const size_t n = (bytes + kPinBytes - 1) / kPinBytes;
    auto issue = [&](size_t i) -> hipError_t {
        const size_t len = std::min(kPinBytes, bytes - i * kPinBytes);
        hipError_t e = hipMemcpyAsync(slot->pin[i & 1], (const char *)src + i * kPinBytes, len, hipMemcpyDeviceToHost, stream);
        return e != hipSuccess ? e : hipEventRecord(slot->pin_ev[i & 1], stream);
    };
    HIP_TRY(issue(0));
    for (size_t i = 0; i < n; ++i) {
        HIP_TRY(hipEventSynchronize(slot->pin_ev[i & 1]));
        if (i + 1 < n) HIP_TRY(issue(i + 1));  // lands in the other buffer while this one is copied out
        const size_t len = std::min(kPinBytes, bytes - i * kPinBytes);
        const size_t part = (len / kCopyThreads + 4095) & ~(size_t)4095;
        char *out = (char *)dst + i * kPinBytes;
        const char *in = slot->pin[i & 1];
        std::thread helpers[kCopyThreads - 1];
        for (int k = 1; k < kCopyThreads; ++k) {
            const size_t a = (size_t)k * part;
            if (a >= len) continue;
            try {
                helpers[k - 1] = std::thread([=] { std::memcpy(out + a, in + a, std::min(part, len - a)); });
            } catch (const std::system_error &) {  // no thread to be had: this one copies the piece itself
                std::memcpy(out + a, in + a, std::min(part, len - a));
            }
        }
        std::memcpy(out, in, std::min(part, len));
        for (auto &h : helpers)
            if (h.joinable()) h.join();
    }
    return RTX_OK;
}

// Host image of the device tables of one scene in one precision.
template <typename R>
struct Tables {
    std::vector<XformRec<R>> sph, cub;
    std::vector<PlaneRec<R>> pln;
    std::vector<CappedRec<R>> cyl, con;
    std::vector<ChainRec<R>> chn;
    std::vector<BoundRec<R>> bnd, cl, grp;
    uint32_t cl_off[5] = { 0, 0, 0, 0, 0 }, grp_off[5] = { 0, 0, 0, 0, 0 };
    std::vector<uint32_t> slot_obj;
    std::vector<uint32_t> slot_flags;
    std::vector<ColdRec<R>> cold;
    // sphere around every bounded object (r = +inf: none, or no bounding-sphere tests)
    double world_c[3] = { 0, 0, 0 }, world_r = std::numeric_limits<double>::infinity();
};

struct Bound {
    double c[3], r;
};

void bounding_sphere(const rtx_scene_desc *d, uint32_t i, double c[3], double *r);

std::vector<Bound> scene_bounds(const rtx_scene_desc *d, uint32_t *n_bounded)
{
    std::vector<Bound> b(d->n_objects);
    *n_bounded = 0;
    for (uint32_t i = 0; i < d->n_objects; ++i) {
        bounding_sphere(d, i, b[i].c, &b[i].r);
        if (d->chain_depth && d->chain_depth[i]) {  // its cached inverse is not the one the ray goes through
            b[i].c[0] = b[i].c[1] = b[i].c[2] = 0.0;
            b[i].r = std::numeric_limits<double>::infinity();
        }
        if (std::isfinite(b[i].r)) ++*n_bounded;
    }
    return b;
}

// Order of the objects of one shape inside its table.  Without the bounding-sphere tests it is
// the World's order.  With them, bounded objects come first in the order of the leaves of a k-d
// partition of their centres (longest axis of the box, cut near the median at a multiple of
// kCluster, or of kCluster^2 while more than that many remain), so that every aligned run of
// kCluster slots, and of kCluster^2 slots, is compact in space; unbounded ones follow.  The order
// inside a table is free: ties between equal t are settled by World index, not by slot.
void kd_order(std::vector<uint32_t> &idx, size_t lo, size_t hi, const std::vector<Bound> &bounds)
{
    const size_t n = hi - lo;
    if (n <= kCluster) return;
    const size_t unit = n > (size_t)kCluster * kCluster ? (size_t)kCluster * kCluster : kCluster;
    const size_t half = ((n / 2 + unit - 1) / unit) * unit;  // 0 < half < n
    double bl[3] = { 1e300, 1e300, 1e300 }, bh[3] = { -1e300, -1e300, -1e300 };
    for (size_t k = lo; k < hi; ++k)
        for (int a = 0; a < 3; ++a) {
            bl[a] = std::min(bl[a], bounds[idx[k]].c[a]);
            bh[a] = std::max(bh[a], bounds[idx[k]].c[a]);
        }
    int axis = 0;
    for (int a = 1; a < 3; ++a)
        if (bh[a] - bl[a] > bh[axis] - bl[axis]) axis = a;
    std::nth_element(idx.begin() + lo, idx.begin() + lo + half, idx.begin() + hi, [&](uint32_t x, uint32_t y) {
        return bounds[x].c[axis] < bounds[y].c[axis] || (bounds[x].c[axis] == bounds[y].c[axis] && x < y);
    });
    kd_order(idx, lo, lo + half, bounds);
    kd_order(idx, lo + half, hi, bounds);
}

std::vector<uint32_t> table_order(const rtx_scene_desc *d, const std::vector<Bound> &bounds, uint32_t shape, bool cull)
{
    std::vector<uint32_t> idx, unbounded;
    for (uint32_t i = 0; i < d->n_objects; ++i)
        if (d->shape_type[i] == shape && !(d->chain_depth && d->chain_depth[i]))
            (!cull || std::isfinite(bounds[i].r) ? idx : unbounded).push_back(i);
    if (!cull) return idx;
    kd_order(idx, 0, idx.size(), bounds);
    idx.insert(idx.end(), unbounded.begin(), unbounded.end());
    return idx;
}

// Sphere that reaches every member's sphere: centred on the mean of the members' centres or on
// the centre of the box around the members' spheres, whichever needs the smaller radius.
Bound enclosing_sphere(const Bound *members, size_t n)
{
    Bound out = { { 0, 0, 0 }, std::numeric_limits<double>::infinity() };
    for (size_t k = 0; k < n; ++k)
        if (!std::isfinite(members[k].r)) return out;
    double mean[3] = { 0, 0, 0 }, lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    for (size_t k = 0; k < n; ++k)
        for (int a = 0; a < 3; ++a) {
            mean[a] += members[k].c[a] / (double)n;
            lo[a] = std::min(lo[a], members[k].c[a] - members[k].r);
            hi[a] = std::max(hi[a], members[k].c[a] + members[k].r);
        }
    const double mid[3] = { 0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2]) };
    const double *centres[2] = { mean, mid };
    for (const double *c : centres) {
        double r = 0.0;
        for (size_t k = 0; k < n; ++k) {
            const double dx = members[k].c[0] - c[0], dy = members[k].c[1] - c[1], dz = members[k].c[2] - c[2];
            r = std::max(r, std::sqrt(dx * dx + dy * dy + dz * dz) * (1.0 + 1e-12) + members[k].r);
        }
        if (r < out.r) {
            out.r = r;
            for (int a = 0; a < 3; ++a) out.c[a] = c[a];
        }
    }
    return out;
}

// Largest singular value of a 3x3 matrix (row-major): square root of the largest eigenvalue of
// A^T A, by the trigonometric solution of the symmetric 3x3 eigenproblem.
double largest_singular_value(const double a[9])
{
    double b[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) b[i][j] = a[i] * a[j] + a[3 + i] * a[3 + j] + a[6 + i] * a[6 + j];
    const double p1 = b[0][1] * b[0][1] + b[0][2] * b[0][2] + b[1][2] * b[1][2];
    double top;
    if (p1 == 0.0) {
        top = std::max(b[0][0], std::max(b[1][1], b[2][2]));
    } else {
        const double q = (b[0][0] + b[1][1] + b[2][2]) / 3.0;
        const double p2 = (b[0][0] - q) * (b[0][0] - q) + (b[1][1] - q) * (b[1][1] - q) + (b[2][2] - q) * (b[2][2] - q) + 2.0 * p1;
        const double p = std::sqrt(p2 / 6.0);
        double c[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) c[i][j] = (b[i][j] - (i == j ? q : 0.0)) / p;
        const double det = c[0][0] * (c[1][1] * c[2][2] - c[1][2] * c[2][1]) - c[0][1] * (c[1][0] * c[2][2] - c[1][2] * c[2][0]) +
                           c[0][2] * (c[1][0] * c[2][1] - c[1][1] * c[2][0]);
        const double r = std::min(1.0, std::max(-1.0, det / 2.0));
        top = q + 2.0 * p * std::cos(std::acos(r) / 3.0);
    }
    return std::sqrt(std::max(top, 0.0));
}

// World-space bounding sphere of object i.  The object's transform is the inverse of the cached
// inverse (affine: x -> A x + b), so an object-space point p lands at A^-1 (p - b).  The sphere
// is centred on the image of (0, y_mid, 0); its radius is the largest singular value of A^-1
// for a sphere and the farthest image of a corner of the enclosing box for a cube or a
// cylinder (the farthest point of the image of a box is the image of a corner).  r = +inf when
// the shape is unbounded or A is not invertible in double precision.
void bounding_sphere(const rtx_scene_desc *d, uint32_t i, double c[3], double *r)
{
    const double inf = std::numeric_limits<double>::infinity();
    c[0] = c[1] = c[2] = 0.0;
    *r = inf;
    double ymid = 0.0, half_y = 1.0;
    const double lo = d->ymin[i], hi = d->ymax[i];
    switch (d->shape_type[i]) {
    case RTX_SPHERE:
    case RTX_CUBE: break;
    case RTX_CYLINDER:
        // the reference's default limits are f64::MIN and f64::MAX (src/shapes/cylinders.rs:8)
        if (!(std::fabs(lo) < 1e100) || !(std::fabs(hi) < 1e100)) return;
        ymid = 0.5 * (lo + hi);
        half_y = 0.5 * std::fabs(hi - lo);
        break;
    // A cone is never bounded: when the ray is parallel to one of its halves the reference
    // keeps the single root -c/(2b) without clipping it to [min, max] (src/shapes/cones.rs:44-52),
    // and its caps accept x^2 + z^2 <= |y| rather than y^2 (cones.rs:103).
    default: return;  // plane, cone
    }
    const double *m = d->inverse + (size_t)i * 16;
    const double a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6], a20 = m[8], a21 = m[9], a22 = m[10];
    const double det = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
    if (!std::isfinite(det) || std::fabs(det) < 1e-300) return;
    const double inv[9] = { (a11 * a22 - a12 * a21) / det, (a02 * a21 - a01 * a22) / det, (a01 * a12 - a02 * a11) / det,
                            (a12 * a20 - a10 * a22) / det, (a00 * a22 - a02 * a20) / det, (a02 * a10 - a00 * a12) / det,
                            (a10 * a21 - a11 * a20) / det, (a01 * a20 - a00 * a21) / det, (a00 * a11 - a01 * a10) / det };
    const double v[3] = { 0.0 - m[3], ymid - m[7], 0.0 - m[11] };
    for (int k = 0; k < 3; ++k) c[k] = inv[k * 3] * v[0] + inv[k * 3 + 1] * v[1] + inv[k * 3 + 2] * v[2];
    double rad = 0.0;
    if (d->shape_type[i] == RTX_SPHERE) {
        rad = largest_singular_value(inv);
    } else {
        for (int corner = 0; corner < 8; ++corner) {
            const double p[3] = { (corner & 1) ? 1.0 : -1.0, (corner & 2) ? half_y : -half_y, (corner & 4) ? 1.0 : -1.0 };
            double q2 = 0.0;
            for (int k = 0; k < 3; ++k) {
                const double q = inv[k * 3] * p[0] + inv[k * 3 + 1] * p[1] + inv[k * 3 + 2] * p[2];
                q2 += q * q;
            }
            if (!(q2 == q2)) return;
            rad = std::max(rad, std::sqrt(q2));
        }
    }
    rad *= 1.0 + 1e-9;
    if (!std::isfinite(rad) || !std::isfinite(c[0]) || !std::isfinite(c[1]) || !std::isfinite(c[2])) {
        c[0] = c[1] = c[2] = 0.0;
        return;
    }
    *r = rad;
}

template <typename R>
void fill_tables(const rtx_scene_desc *d, const std::vector<uint8_t> &container, const std::vector<Bound> &bounds, bool cull,
                 Tables<R> &t)
{
    const uint32_t n = d->n_objects;
    t.cold.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        ColdRec<R> &c = t.cold[i];
        std::memset(&c, 0, sizeof c);
        const double *inv = d->inverse + (size_t)i * 16;
        const double *invt = d->inverse_transposed + (size_t)i * 16;
        const double *pinv = d->pattern_inverse + (size_t)i * 16;
        for (int k = 0; k < 12; ++k) c.inv[k] = (R)inv[k];
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) c.inv_t[r * 3 + k] = (R)invt[r * 4 + k];
        c.ymin = (R)d->ymin[i];
        c.ymax = (R)d->ymax[i];
        for (int k = 0; k < 3; ++k) {
            c.color[k] = (R)d->color[(size_t)i * 3 + k];
            c.c1[k] = (R)d->pattern_color1[(size_t)i * 3 + k];
            c.c2[k] = (R)d->pattern_color2[(size_t)i * 3 + k];
        }
        c.ambient = (R)d->ambient[i];
        c.diffuse = (R)d->diffuse[i];
        c.specular = (R)d->specular[i];
        c.shininess = (R)d->shininess[i];
        c.reflective = (R)d->reflective[i];
        c.transparency = (R)d->transparency[i];
        c.refractive_index = (R)d->refractive_index[i];
        for (int k = 0; k < 12; ++k) c.pinv[k] = (R)pinv[k];
        c.type = d->shape_type[i];
        c.flags = (d->closed[i] ? OF_CLOSED : 0u) | (d->casts_shadow[i] ? OF_CASTS_SHADOW : 0u) |
                  (container[i] ? OF_CONTAINER : 0u);
        c.pattern_type = d->pattern_type[i];
    }
    // intersection tables: one per shape
    // inflation of a sphere stored as R: a relative part for the rounding of the test, an
    // absolute part for the rounding of the centre
    const double rel = sizeof(R) == 8 ? 1e-6 : 1e-3, ulp = sizeof(R) == 8 ? 1e-14 : 1e-6;
    auto store = [&](const double c[3], double r) {
        BoundRec<R> b;
        const double cmax = std::max(std::fabs(c[0]), std::max(std::fabs(c[1]), std::fabs(c[2])));
        for (int k = 0; k < 3; ++k) b.c[k] = (R)c[k];
        b.r = std::isfinite(r) ? (R)(r * (1.0 + rel) + cmax * ulp) : (R)std::numeric_limits<double>::infinity();
        return b;
    };
    for (uint32_t shape = RTX_SPHERE; shape <= RTX_CONE; ++shape) {
        const std::vector<uint32_t> order = table_order(d, bounds, shape, cull);
        t.cl_off[shape] = (uint32_t)t.cl.size();
        t.grp_off[shape] = (uint32_t)t.grp.size();
        if (cull) {
            std::vector<Bound> members, clusters;
            for (uint32_t i : order) members.push_back(bounds[i]);
            for (size_t k0 = 0; k0 < members.size(); k0 += kCluster)
                clusters.push_back(enclosing_sphere(members.data() + k0, std::min<size_t>(kCluster, members.size() - k0)));
            for (const Bound &c : clusters) t.cl.push_back(store(c.c, c.r));
            for (size_t k0 = 0; k0 < clusters.size(); k0 += kCluster) {
                const Bound g = enclosing_sphere(clusters.data() + k0, std::min<size_t>(kCluster, clusters.size() - k0));
                t.grp.push_back(store(g.c, g.r));
            }
        }
        for (uint32_t i : order) {
            const double *inv = d->inverse + (size_t)i * 16;
            t.slot_obj.push_back(i);
            // an inverse whose upper-left 3x3 is diagonal: the kernels take a ray into this object's space with
            // a multiply per component (zeros of either sign count; a plane reads row 1 only)
            const bool diagonal = shape == RTX_PLANE ? (inv[4] == 0.0 && inv[6] == 0.0)
                                                     : (inv[1] == 0.0 && inv[2] == 0.0 && inv[4] == 0.0 && inv[6] == 0.0 &&
                                                        inv[8] == 0.0 && inv[9] == 0.0);
            t.slot_flags.push_back(diagonal ? SLOT_DIAGONAL : 0u);
            t.bnd.push_back(store(bounds[i].c, bounds[i].r));
            if (shape == RTX_SPHERE || shape == RTX_CUBE) {
                XformRec<R> r;
                for (int k = 0; k < 12; ++k) r.m[k] = (R)inv[k];
                (shape == RTX_SPHERE ? t.sph : t.cub).push_back(r);
            } else if (shape == RTX_PLANE) {
                PlaneRec<R> r;
                for (int k = 0; k < 4; ++k) r.m[k] = (R)inv[4 + k];
                t.pln.push_back(r);
            } else {
                CappedRec<R> r;
                std::memset(&r, 0, sizeof r);
                for (int k = 0; k < 12; ++k) r.m[k] = (R)inv[k];
                r.ymin = (R)d->ymin[i];
                r.ymax = (R)d->ymax[i];
                r.closed = d->closed[i] ? 1u : 0u;
                (shape == RTX_CYLINDER ? t.cyl : t.con).push_back(r);
            }
        }
    }
    // objects inside groups: last in slot order, World order among themselves
    for (uint32_t i = 0; d->chain_depth && i < n; ++i) {
        const uint32_t depth = d->chain_depth[i];
        if (!depth) continue;
        ChainRec<R> r;
        std::memset(&r, 0, sizeof r);
        for (uint32_t k = 0; k <= depth; ++k) {
            const double *m = d->chain_inverse + ((size_t)d->chain_first[i] + k) * 16;
            for (int e = 0; e < 12; ++e) r.m[k][e] = (R)m[e];
        }
        r.ymin = (R)d->ymin[i];
        r.ymax = (R)d->ymax[i];
        r.type = d->shape_type[i];
        r.closed = d->closed[i] ? 1u : 0u;
        r.depth = depth;
        t.chn.push_back(r);
        t.slot_obj.push_back(i);
        t.slot_flags.push_back(0u);
        t.bnd.push_back(store(bounds[i].c, bounds[i].r));
    }
    // The world sphere: around every object that has a finite sphere, with room for the inflation above.
    std::vector<Bound> finite;
    for (uint32_t i = 0; cull && i < n; ++i)
        if (std::isfinite(bounds[i].r)) finite.push_back(bounds[i]);
    if (!finite.empty()) {
        const Bound w = enclosing_sphere(finite.data(), finite.size());
        if (std::isfinite(w.r)) {
            const double cmax = std::max(std::fabs(w.c[0]), std::max(std::fabs(w.c[1]), std::fabs(w.c[2])));
            for (int k = 0; k < 3; ++k) t.world_c[k] = w.c[k];
            t.world_r = w.r * (1.0 + 4.0 * rel) + cmax * 4.0 * ulp + 1e-3;
        }
    }
    // the walk fetches spheres four at a time: three entries of padding after the last
    const double zero[3] = { 0.0, 0.0, 0.0 };
    for (int k = 0; k < 3; ++k) {
        t.bnd.push_back(store(zero, std::numeric_limits<double>::infinity()));
        t.cl.push_back(store(zero, std::numeric_limits<double>::infinity()));
    }
}

// The packed copy of the sphere hierarchy (QuantNode): the spheres of `t` (objects by slot, clusters,
// groups, padding included) on a grid of 65 536 steps per axis over the box of their centres.  A packed
// sphere holds the stored one: its radius is rounded up and widened by what the centre can have moved
// (half a step per axis), by what single-precision arithmetic on the unpacked numbers loses, and by what
// a ray loses when its double-precision origin and direction are rounded to single precision.  Empty
// (nodes.size() == 0) when the scene has no hierarchy or it does not fit a workgroup's LDS.
struct QuantTree {
    std::vector<QuantNode> nodes;
    float base[3] = { 0.f, 0.f, 0.f }, step = 1.f;
};
QuantTree quantize_tree(const Tables<double> &t)
{
    QuantTree q;
    const size_t total = t.bnd.size() + t.cl.size() + t.grp.size();
    if (t.grp.empty() || total + 16 > kLaneTreeMaxNodes) return q;  // (the walk reads up to 15 entries past a table's last)
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    bool any = false;
    for (const auto *v : { &t.bnd, &t.cl, &t.grp })
        for (const BoundRec<double> &b : *v)
            if (std::isfinite(b.r)) {
                any = true;
                for (int a = 0; a < 3; ++a) {
                    lo[a] = std::min(lo[a], b.c[a]);
                    hi[a] = std::max(hi[a], b.c[a]);
                }
            }
    if (!any) return q;
    double extent = 0.0, maxabs = 0.0;
    for (int a = 0; a < 3; ++a) {
        extent = std::max(extent, hi[a] - lo[a]);
        maxabs = std::max(maxabs, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
    }
    const double world = maxabs + extent + (std::isfinite(t.world_r) ? 2.0 * t.world_r : 0.0);
    const double step = std::max(extent / 65535.0, std::max(world * 1e-7, 1e-30));
    if (!std::isfinite(step) || !std::isfinite(world) || world > 1e30) return q;
    q.step = (float)step;
    const double stepf = (double)q.step;
    for (int a = 0; a < 3; ++a) q.base[a] = (float)lo[a];
    // the centre moves by half a step per axis (sqrt(3)/2 of a step) plus the rounding of base and of the
    // unpacking; rays rounded to single precision move by a few units in the last place of the coordinates
    const double margin = 0.9 * stepf + 64.0 * 1.2e-7 * world;
    q.nodes.reserve(total);
    for (const auto *v : { &t.bnd, &t.cl, &t.grp })
        for (const BoundRec<double> &b : *v) {
            QuantNode n = { { 0, 0, 0 }, 0xffffu };
            if (std::isfinite(b.r)) {
                bool inside = true;
                for (int a = 0; a < 3; ++a) {
                    const double g = std::nearbyint((b.c[a] - (double)q.base[a]) / stepf);
                    if (!(g >= 0.0 && g <= 65535.0)) inside = false;
                    else n.c[a] = (uint16_t)g;
                }
                const double r = std::ceil((b.r * (1.0 + 2e-3) + margin) / stepf) + 1.0;
                if (inside && r < 65535.0) n.r = (uint16_t)r;
                else n.c[0] = n.c[1] = n.c[2] = 0;
            }
            q.nodes.push_back(n);
        }
    return q;
}

// One device allocation per scene and precision holding all tables back to back.
template <typename R>
int upload_tables(const Tables<R> &t, const QuantTree &qt, uint32_t sf, const rtx_scene_desc *d, void **blob, SceneDev<R> *dev)
{
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_sph = pad(t.sph.size() * sizeof(XformRec<R>)), b_pln = pad(t.pln.size() * sizeof(PlaneRec<R>)),
                 b_cub = pad(t.cub.size() * sizeof(XformRec<R>)), b_cyl = pad(t.cyl.size() * sizeof(CappedRec<R>)),
                 b_con = pad(t.con.size() * sizeof(CappedRec<R>)), b_slot = pad(t.slot_obj.size() * sizeof(uint32_t)),
                 b_flags = pad(t.slot_flags.size() * sizeof(uint32_t)),
                 b_cold = pad(t.cold.size() * sizeof(ColdRec<R>)), b_bnd = pad(t.bnd.size() * sizeof(BoundRec<R>)),
                 b_cl = pad(t.cl.size() * sizeof(BoundRec<R>)), b_grp = pad(t.grp.size() * sizeof(BoundRec<R>)),
                 b_chn = pad(t.chn.size() * sizeof(ChainRec<R>)), b_q = pad(qt.nodes.size() * sizeof(QuantNode));
    const size_t total = b_sph + b_pln + b_cub + b_cyl + b_con + b_slot + b_flags + b_cold + b_bnd + b_cl + b_grp + b_chn + b_q + 256;
    std::vector<char> host(total, 0);
    size_t off = 0;
    auto put = [&](const void *src, size_t bytes, size_t padded) {
        if (bytes) std::memcpy(host.data() + off, src, bytes);
        const size_t at = off;
        off += padded;
        return at;
    };
    const size_t o_sph = put(t.sph.data(), t.sph.size() * sizeof(XformRec<R>), b_sph);
    const size_t o_pln = put(t.pln.data(), t.pln.size() * sizeof(PlaneRec<R>), b_pln);
    const size_t o_cub = put(t.cub.data(), t.cub.size() * sizeof(XformRec<R>), b_cub);
    const size_t o_cyl = put(t.cyl.data(), t.cyl.size() * sizeof(CappedRec<R>), b_cyl);
    const size_t o_con = put(t.con.data(), t.con.size() * sizeof(CappedRec<R>), b_con);
    const size_t o_slot = put(t.slot_obj.data(), t.slot_obj.size() * sizeof(uint32_t), b_slot);
    const size_t o_flags = put(t.slot_flags.data(), t.slot_flags.size() * sizeof(uint32_t), b_flags);
    const size_t o_cold = put(t.cold.data(), t.cold.size() * sizeof(ColdRec<R>), b_cold);
    const size_t o_bnd = put(t.bnd.data(), t.bnd.size() * sizeof(BoundRec<R>), b_bnd);
    const size_t o_cl = put(t.cl.data(), t.cl.size() * sizeof(BoundRec<R>), b_cl);
    const size_t o_grp = put(t.grp.data(), t.grp.size() * sizeof(BoundRec<R>), b_grp);
    const size_t o_chn = put(t.chn.data(), t.chn.size() * sizeof(ChainRec<R>), b_chn);
    const size_t o_q = put(qt.nodes.data(), qt.nodes.size() * sizeof(QuantNode), b_q);
    HIP_TRY(hipMalloc(blob, total));
    HIP_TRY(hipMemcpy(*blob, host.data(), total, hipMemcpyHostToDevice));
    char *base = (char *)*blob;
    std::memset(dev, 0, sizeof *dev);
    dev->sph = (const XformRec<R> *)(base + o_sph);
    dev->pln = (const PlaneRec<R> *)(base + o_pln);
    dev->cub = (const XformRec<R> *)(base + o_cub);
    dev->cyl = (const CappedRec<R> *)(base + o_cyl);
    dev->con = (const CappedRec<R> *)(base + o_con);
    dev->slot_obj = (const uint32_t *)(base + o_slot);
    dev->slot_flags = (const uint32_t *)(base + o_flags);
    dev->cold = (const ColdRec<R> *)(base + o_cold);
    dev->bnd = (const BoundRec<R> *)(base + o_bnd);
    dev->cl = (const BoundRec<R> *)(base + o_cl);
    dev->grp = (const BoundRec<R> *)(base + o_grp);
    dev->chn = (const ChainRec<R> *)(base + o_chn);
    dev->q_nodes = qt.nodes.empty() ? nullptr : (const QuantNode *)(base + o_q);
    dev->q_n = (uint32_t)qt.nodes.size();
    dev->q_lds = 0u;
    for (int k = 0; k < 3; ++k) dev->q_base[k] = qt.base[k];
    dev->q_step = qt.step;
    dev->n_chn = (uint32_t)t.chn.size();
    dev->n_grp_total = (uint32_t)t.grp.size();
    dev->n_bnd_total = (uint32_t)t.bnd.size();
    dev->n_cl_total = (uint32_t)t.cl.size();
    for (int k = 0; k < 5; ++k) {
        dev->cl_off[k] = t.cl_off[k];
        dev->grp_off[k] = t.grp_off[k];
    }
    dev->n_sph = (uint32_t)t.sph.size();
    dev->n_pln = (uint32_t)t.pln.size();
    dev->n_cub = (uint32_t)t.cub.size();
    dev->n_cyl = (uint32_t)t.cyl.size();
    dev->n_con = (uint32_t)t.con.size();
    dev->n = d->n_objects;
    dev->flags = sf;
    for (int k = 0; k < 3; ++k) dev->world_c[k] = (R)t.world_c[k];
    dev->world_r = std::isfinite(t.world_r) ? (R)t.world_r : std::numeric_limits<R>::infinity();
    if (d->n_lights > 0 && d->light_position && d->light_intensity)
        for (int k = 0; k < 3; ++k) {
            dev->light_pos[k] = (R)d->light_position[k];
            dev->light_int[k] = (R)d->light_intensity[k];
        }
    return RTX_OK;
}

template <typename R>
CameraDev<R> make_camera(const rtx_camera_desc *cam)
{
    CameraDev<R> c;
    std::memset(&c, 0, sizeof c);
    const double *m = cam->inverse_transform;
    for (int k = 0; k < 12; ++k) c.inv[k] = (R)m[k];
    // inverse_transform * point(0, 0, 0), row by row as src/matrices/matrix4.rs:172-185 does it
    for (int i = 0; i < 3; ++i) {
        double val = 0.0;
        const double other[4] = { 0.0, 0.0, 0.0, 1.0 };
        for (int a = 0; a < 4; ++a) val += m[i * 4 + a] * other[a];
        c.origin[i] = (R)val;
    }
    c.pixel_size = (R)cam->pixel_size;
    c.half_width = (R)cam->half_width;
    c.half_height = (R)cam->half_height;
    c.hsize = cam->hsize;
    c.vsize = cam->vsize;
    c.inv_hsize = 1.0 / (double)cam->hsize;
    return c;
}

template <typename R> const SceneDev<R> &scene_dev(const rtx_scene *s);
template <> const SceneDev<double> &scene_dev<double>(const rtx_scene *s) { return s->dev64; }
template <> const SceneDev<float> &scene_dev<float>(const rtx_scene *s) { return s->dev32; }

uint32_t grid_for(const rtx_context *ctx, uint64_t n)
{
    uint64_t blocks = (n + kBlock - 1) / kBlock;
    uint64_t cap = (uint64_t)ctx->cu_count * ctx->blocks_per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (uint32_t)blocks;
}

// Most children one color_at node can queue: a reflection ray if any object reflects, a
// refraction ray if any object is transparent.
uint64_t children_per_hit(uint32_t sf) { return ((sf & SF_REFLECTIVE) ? 1u : 0u) + ((sf & SF_TRANSPARENT) ? 1u : 0u); }

struct FramePlan {
    uint32_t levels = 0;        // bounce launches per chunk
    uint32_t n_sub = 1;         // sub-queues per queue (power of two)
    uint64_t chunk_units = 0;   // rows (camera renders) or rays (batches) per chunk
    uint64_t cap_sub[kMaxLevels + 2] = {};  // entries per sub-queue of the queue that feeds level L
    uint64_t odd_bytes = 0, even_bytes = 0;  // the two ping-pong buffers
    size_t bytes = 0;
};

// Queue sizes for chunks of `units` rows / rays.  Workgroup b of the primary kernel appends to
// sub-queue b mod n_sub and each of its rays (one per thread) queues at most g children, which
// bounds the level-1 sub-queues; a bounce launch feeds every output sub-queue from exactly
// ceil(cap_sub / bounce_feed) blocks, which bounds the next level the same way.
// Scenes with regional levels: a camera workgroup appends to the sub-queues of its screen region, one per
// kind of child, each of which holds at most one ray per pixel of the region; the children of a ray stay
// in the sub-queues of its region, so a sub-queue of the next level holds at most what all sub-queues of
// the region hold now.
template <typename R>
void size_queues(uint32_t sf, uint32_t levels, uint32_t hsize, uint64_t units, FramePlan *p)
{
    const uint64_t g = children_per_hit(sf);
    const bool regional = regional_levels(sf);
    const uint64_t blocks0 = hsize ? camera_blocks<R>(sf, hsize, units) : (units + kRayBlock<R> - 1) / kRayBlock<R>;
    uint32_t n_sub = 1;
    const uint32_t most_sub = regional ? kListMaxSub : 256u;
    while (n_sub < most_sub && (uint64_t)n_sub * 2 * 8 <= blocks0) n_sub *= 2;
    const uint32_t kinds = layout_kinds(sf);  // sub-queues per region: one per kind of child the kernels can queue
    if (regional && n_sub < kinds) n_sub = kinds;
    p->levels = levels;
    p->n_sub = n_sub;
    p->chunk_units = units;
    p->odd_bytes = p->even_bytes = 0;
    for (auto &c : p->cap_sub) c = 0;
    // rays per primary workgroup: kRayBlock of a ray batch, kCamBlock pixels of a frame, of whose
    // 8x8 tiles only `units` rows exist when the chunk is lower than a tile
    const uint64_t per_block0 = !hsize ? (uint64_t)kRayBlock<R> : ((sf & SF_CULL) && units < 8) ? (kCamBlock<R> / 8) * units : (uint64_t)kCamBlock<R>;
    uint64_t cap = ((blocks0 + n_sub - 1) / n_sub) * per_block0 * g;
    if (regional) {
        const uint32_t n_regions = n_sub / kinds;
        if (hsize) {
            uint32_t nx, cols, rows;
            region_split(camera_blocks_per_row<R>(sf, hsize), camera_piece_rows(sf, units), n_regions, &nx, &cols, &rows);
            cap = (uint64_t)cols * rows * per_block0;
        } else {
            cap = ((blocks0 + n_regions - 1) / n_regions) * per_block0;  // workgroup b of a ray batch: region b mod n_regions
        }
    }
    for (uint32_t L = 1; L <= levels; ++L) {
        p->cap_sub[L] = cap;
        uint64_t &tot = (L & 1) ? p->odd_bytes : p->even_bytes;
        tot = std::max(tot, queue_buffer_bytes<R>(sf, cap * n_sub, n_sub));
        if (regional) {
            cap *= kinds;
        } else {
            // (an output sub-queue is fed by one block per block index of the input; a block holds at most
            // bounce_feed rays, and no more than its whole sub-queue)
            const uint64_t piece = bounce_feed<R>(sf);
            cap = ((cap + piece - 1) / piece) * std::min<uint64_t>(piece, cap) * g;
        }
    }
    p->bytes = (size_t)(p->odd_bytes + p->even_bytes);
}

// Split the frame into chunks of whole rows (or of rays, hsize = 0) so that both ping-pong
// queues, sized for the worst case, fit the workspace limit.
template <typename R>
int plan_frame(size_t limit, uint32_t sf, uint32_t max_depth, uint32_t hsize, uint64_t n_units, FramePlan *p)
{
    if (max_depth >= kMaxLevels) return fail(RTX_ERR_INVALID, "max_depth %u is not supported (limit %u)", max_depth, kMaxLevels - 1);
    const bool secondary = children_per_hit(sf) != 0 && max_depth > 0;
    const uint32_t levels = secondary ? max_depth : 0;
    // one launch covers at most 2^31 - 256 primary rays
    // (and, of a frame, at most the rows a camera launch can number)
    const uint64_t max_units = hsize ? std::min<uint64_t>(std::max<uint64_t>(1, 0x7fffff00ull / hsize), kCameraMaxRows) : 0x7fffff00ull;
    auto fits = [&](uint64_t units) {
        size_queues<R>(sf, levels, hsize, units, p);
        if (p->bytes > limit) return false;
        for (uint32_t L = 1; L <= levels; ++L)
            if (p->cap_sub[L] > 0x7fffff00ull) return false;  // sub-queue slots are indexed with 32 bits
        for (uint32_t L = 1; L <= levels; ++L)
            if (regional_levels(sf) && p->cap_sub[L] * p->n_sub > 0xffffff00ull) return false;  // and so are the entries of a regional queue
        return true;
    };
    uint64_t hi = std::min<uint64_t>(std::max<uint64_t>(n_units, 1), max_units);
    if (!fits(hi)) {
        if (!fits(1)) return fail(RTX_ERR_INVALID, "workspace limit too small for one row of this frame");
        uint64_t lo = 1;  // fits(lo) holds, fits(hi) does not
        while (hi - lo > 1) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if (fits(mid)) lo = mid; else hi = mid;
        }
        (void)fits(lo);
    }
    return RTX_OK;
}

// A queue overflow drops rays: the frame it happened in is wrong.  The kernels report it through
// a word of pinned host memory, which every later call on the context looks at.
int check_overflow(rtx_context *ctx)
{
    if (ctx->overflow_host && __atomic_load_n(ctx->overflow_host, __ATOMIC_ACQUIRE) != 0u) {
        __atomic_store_n(ctx->overflow_host, 0u, __ATOMIC_RELEASE);
        return fail(RTX_ERR_DEVICE, "secondary-ray queue overflow in an earlier frame of this context (internal sizing error): "
                                    "that frame is incomplete");
    }
    return RTX_OK;
}

// The bounce levels of one chunk of a measured frame: the level-1 queue (ws.queues, plan.cap_sub[1]
// entries per sub-queue) has been filled by the primary launch already queued on `stream`.  Before
// each level the lengths of its sub-queues are read back; block b of sub-queue y (bounce_block rays,
// the last one fewer) sends its children to sub-queue (b + y) mod n_sub of the next queue
// (k_bounce, k_bounce_tree), which is therefore sized for the largest sum of such blocks times the children a ray
// can have; in a scene with regional levels the children stay in the sub-queues of their region, and the next
// queue gets room for twice the rays of this one, sub-queue by sub-queue (k_list_blocks lays it out).  A level
// without rays ends the chunk.
template <typename R>
int enqueue_measured_levels(rtx_context *ctx, const rtx_scene *scene, Workspace &ws, hipStream_t stream, const FramePlan &plan,
                            uint32_t *counts, uint32_t max_depth, R *fb, uint32_t *launches)
{
    const uint32_t n_sub = plan.n_sub;
    const uint64_t g = children_per_hit(scene->sf);
    const bool regional = regional_levels(scene->sf);  // the children of a ray stay in the sub-queues of its region
    if (!ws.host_counts) {
        HIP_TRY(hipHostMalloc((void **)&ws.host_counts, (kListMaxSub + 64) * sizeof(uint32_t), hipHostMallocMapped));
        ws.host_counts[0] = 0u;
        ws.publish_seq = 0u;
        HIP_TRY(hipHostGetDevicePointer((void **)&ws.host_counts_dev, ws.host_counts, 0));
    }
    uint64_t cap_in = plan.cap_sub[1];
    const bool tree = (scene->sf & SF_LANE_TREE) != 0;
    std::vector<uint64_t> feed(n_sub);
    // Regional queues: the level-1 queue has n_sub sub-queues of one capacity; every deeper queue is laid out by
    // k_list_blocks from the lengths of the queue before it (sub-queues back to back, each with room for what its
    // region holds), and holds `kinds` times the rays of that queue.
    uint64_t entries_in = cap_in * n_sub, entries_out = 0;
    bool var_in = false;
    for (uint32_t L = 1; L <= max_depth; ++L) {
        uint32_t *level_counts = counts + (size_t)L * n_sub * kCountStride;
        // A one-workgroup kernel behind the level's producer writes the lengths into mapped host memory and
        // then a sequence number, which this thread polls for: about half the latency of a copy and a
        // stream synchronisation.  The stream is queried now and then so that a failed launch ends the wait.
        const uint32_t seq = ++ws.publish_seq;
        HIP_TRY(launch_publish_counts(stream, level_counts, n_sub, kCountStride, ws.host_counts_dev, seq));
        for (uint32_t spins = 0; __atomic_load_n(ws.host_counts, __ATOMIC_ACQUIRE) != seq; ++spins) {
            if ((spins & 0xfffu) == 0xfffu) {
                const hipError_t q = hipStreamQuery(stream);
                if (q == hipSuccess) {
                    if (__atomic_load_n(ws.host_counts, __ATOMIC_ACQUIRE) == seq) break;
                    return fail(RTX_ERR_DEVICE, "the queue lengths of level %u never arrived", L);
                }
                if (q != hipErrorNotReady) return fail(RTX_ERR_DEVICE, "waiting for level %u: %s", L, hipGetErrorString(q));
            }
        }
        uint64_t longest = 0, cap_out = 0, running_region = 0, cap_region = 0, rays_in = 0;
        // feed[s]: rays the blocks that send to output sub-queue s hold.  Sub-queue y sends its block b
        // to (b + y) mod n_sub: every full turn of n_sub blocks adds a block to each s, the blocks
        // left over go to a run of sub-queues that starts at y (kept as differences, summed below).
        std::fill(feed.begin(), feed.end(), 0);
        uint64_t everywhere = 0;
        auto add_run = [&](uint32_t first, uint64_t count, uint64_t rays) {  // count < n_sub sub-queues from `first` on, cyclically
            if (count == 0) return;
            const uint64_t end = first + count;
            feed[first] += rays;
            if (end < n_sub) {
                feed[end] -= rays;
            } else {
                feed[0] += rays;
                if (end - n_sub < n_sub) feed[end - n_sub] -= rays;
            }
        };
        for (uint32_t y = 0; y < n_sub; ++y) {
            const uint64_t n = std::min<uint64_t>(ws.host_counts[1 + y], cap_in);
            longest = std::max(longest, n);
            rays_in += n;
            if (regional) {  // what the sub-queues of this one's region hold together
                if (y % layout_kinds(scene->sf) == 0) running_region = 0;
                running_region += n;
                cap_region = std::max(cap_region, running_region);
                continue;
            }
            // rays per block: bounce_feed lane slots, of which k_bounce_tree gives a ray one or 64
            const uint64_t piece = tree ? bounce_feed<R>(scene->sf) / tree_lanes_per_ray(scene->sf, (uint32_t)n, (uint32_t)cap_in, n_sub)
                                        : bounce_feed<R>(scene->sf);
            const uint64_t full = n / piece, rest = n % piece;
            everywhere += (full / n_sub) * piece;
            add_run(y, full % n_sub, piece);
            if (rest) add_run((uint32_t)((full + y) & (n_sub - 1)), 1, rest);
        }
        uint64_t running = 0;
        for (uint32_t q = 0; q < n_sub; ++q) {
            running += feed[q];  // differences wrap around as unsigned numbers and cancel
            feed[q] = (running + everywhere) * g;
        }
        if (longest == 0) break;
        if (L < max_depth) {
            for (uint64_t f : feed) cap_out = std::max(cap_out, f);
            if (regional) cap_out = cap_region;
            entries_out = regional ? rays_in * layout_kinds(scene->sf) : cap_out * n_sub;
            if (cap_out > 0x7fffff00ull || entries_out > 0xffffff00ull) return fail(RTX_ERR_INVALID, "level %u of this frame has too many rays for one launch; render fewer rows per call", L + 1);
            void **buf = ((L + 1) & 1) ? &ws.queues : &ws.queues_even;
            size_t *have = ((L + 1) & 1) ? &ws.queue_bytes : &ws.queue_even_bytes;
            const size_t need = (size_t)queue_buffer_bytes<R>(scene->sf, entries_out, n_sub);
            if (need > *have) {
                // the buffer holds the queue this level's input was made from: nothing reads it any more
                if (*buf) HIP_TRY(hipFree(*buf));
                *buf = nullptr;
                *have = 0;
                if (hipMalloc(buf, need) != hipSuccess) {
                    (void)hipGetLastError();
                    *buf = nullptr;
                    trim_idle_queues(ctx, &ws);
                    if (hipMalloc(buf, need) != hipSuccess) {
                        (void)hipGetLastError();
                        *buf = nullptr;
                        return fail(RTX_ERR_DEVICE, "out of device memory for the rays of level %u (%zu bytes); render fewer rows per call", L + 1, need);
                    }
                }
                *have = need;
            }
        }
        auto queue_of = [&](uint32_t level, uint64_t cap, uint64_t entries, bool var) {
            QueueDev<R> q;
            q.rec = (R *)((level & 1) ? ws.queues : ws.queues_even);
            q.count = counts + (size_t)level * n_sub * kCountStride;
            q.cap_sub = (uint32_t)cap;
            q.sub_mask = n_sub - 1;
            q.overflow = ctx->overflow_dev;
            queue_list_plane<R>(scene->sf, q, entries, var);
            return q;
        };
        // the launch covers the longest sub-queue that exists, not the capacity
        QueueDev<R> in_q = queue_of(L, cap_in, entries_in, var_in);
        const QueueDev<R> out_q = L < max_depth ? queue_of(L + 1, cap_out, entries_out, regional) : queue_of(L + 1, 0, 0, false);
        HIP_TRY(launch_bounce<R>(scene->sf, stream, scene_dev<R>(scene), in_q, out_q, max_depth - L, L, fb,
                                 ws.shards, tree ? ctx->bounce_blocks_lane : ctx->bounce_blocks, (uint32_t)longest));
        ++*launches;
        // (the list of the next queue, and with it the layout of the queue after that)
        if (L < max_depth)
            HIP_TRY(launch_list_blocks<R>(stream, scene->sf, out_q, L + 1 < max_depth ? counts + (size_t)(L + 2) * n_sub * kCountStride : nullptr));
        cap_in = cap_out;
        entries_in = entries_out;
        var_in = regional;
    }
    return RTX_OK;
}

// Enqueue every kernel of one frame (or ray batch) on `stream`.  Camera renders cover the first
// n_units local rows of `map`; batches cover n_units rays of d_rays.  format 0: d_fb holds three
// values of R per pixel; otherwise it is an 8-bit image written by the primary kernel itself,
// which only a frame without secondary rays can be.  `ev` (four events) is only used when timed.
template <typename R>
int enqueue_frame(rtx_context *ctx, const rtx_scene *scene, Workspace &ws, hipEvent_t *ev, hipStream_t stream,
                  const CameraDev<R> *cam, const double *d_rays, const RowMap &map, uint64_t n_units, uint32_t max_depth, void *d_fb,
                  uint32_t format, rtx_progress_fn progress, void *user, bool timed, uint32_t *n_chunks_out, uint32_t *n_launch_out,
                  bool may_wait = false)
{
    FramePlan plan;
    const uint32_t W = cam ? cam->hsize : 0;
    // Measured frames.  The worst case doubles per level in a scene that both reflects and refracts
    // (26 GB for a 4K depth-5 frame) while the rays that exist do not (under 2 GB there).  A call
    // that waits for its frame anyway can afford to look: it reads the lengths of a level's
    // sub-queues back before it launches that level and sizes the next queue for what those rays
    // can produce at most, at the price of one host round trip per level (about 11 us each: the 4K
    // glass frame goes from 2.06 to 2.10 ms in f64 and from 1.16 to 1.22 ms in f32).
    // The first level is planned alone, for the worst case of the primary rays.
    bool measured = false;
    // (of the scenes with the sphere hierarchy only those whose bounce levels run k_bounce_tree: the host must know
    // how the kernel deals out a sub-queue, and k_bounce's choice of lanes per ray is not repeated here)
    if (may_wait && !ws.fixed && children_per_hit(scene->sf) == 2 && ((scene->sf & SF_CULL) == 0 || (scene->sf & SF_LANE_TREE) != 0) && max_depth >= 2 &&
        max_depth < kMaxLevels && ctx->debug_queue_divisor <= 1) {
        FramePlan whole;
        size_queues<R>(scene->sf, max_depth, W, n_units, &whole);
        measured = whole.bytes > ctx->measured_above;
    }
    int rc = plan_frame<R>(ws.fixed ? ws.queue_bytes : ctx->workspace_limit, scene->sf, measured ? 1u : max_depth, W, n_units, &plan);
    if (rc) return rc;
    if (measured) {
        plan.levels = max_depth;
        plan.even_bytes = 0;
        plan.bytes = (size_t)plan.odd_bytes;
    }
    if (!ws.fixed && plan.bytes > ws.queue_bytes) {
        // Not enough memory for the queues as planned: first take back what idle slots hold, then
        // plan smaller chunks (half the limit at a time) before giving up.
        size_t limit = ctx->workspace_limit;
        for (int attempt = 0;; ++attempt) {
            if (ws.queues) (void)hipFree(ws.queues);
            ws.queues = nullptr;
            ws.queue_bytes = 0;
            if (hipMalloc(&ws.queues, plan.bytes) == hipSuccess) {
                ws.queue_bytes = plan.bytes;
                break;
            }
            (void)hipGetLastError();
            ws.queues = nullptr;
            if (attempt == 0) {
                trim_idle_queues(ctx, &ws);
                continue;
            }
            limit /= 2;
            if (limit < ((size_t)64 << 20)) return fail(RTX_ERR_DEVICE, "out of device memory for the secondary-ray queues (%zu bytes)", plan.bytes);
            // (a measured frame plans its first level alone, as above: the deeper ones are sized as they come)
            rc = plan_frame<R>(limit, scene->sf, measured ? 1u : max_depth, W, n_units, &plan);
            if (rc) return rc;
            if (measured) {
                plan.levels = max_depth;
                plan.even_bytes = 0;
                plan.bytes = (size_t)plan.odd_bytes;
            }
        }
    }
    if (format != 0u && plan.levels != 0u) return fail(RTX_ERR_INVALID, "internal: 8-bit output straight from a frame with secondary rays");
    if (ctx->debug_queue_divisor > 1)
        for (auto &c : plan.cap_sub) c = c ? std::max<uint64_t>(1, c / ctx->debug_queue_divisor) : 0;
    const uint32_t n_chunks = (uint32_t)((n_units + plan.chunk_units - 1) / plan.chunk_units);
    const size_t counts_per_chunk = (size_t)(kMaxLevels + 1) * plan.n_sub * kCountStride;
    const size_t counts_need = (size_t)std::max<uint32_t>(n_chunks, 1) * counts_per_chunk;
    if (ws.fixed) {
        if (plan.bytes > ws.queue_bytes)
            return fail(RTX_ERR_INVALID, "frame needs %zu bytes of ray queues; the fixed workspace holds %zu", plan.bytes, ws.queue_bytes);
        if (counts_need > ws.counts_n)
            return fail(RTX_ERR_INVALID, "frame too large for the fixed workspace: %u row chunks need %zu counter words, it holds %zu",
                        n_chunks, counts_need, ws.counts_n);
    } else {
        rc = grow(&ws.queues, &ws.queue_bytes, plan.bytes);
        if (rc) return rc;
        if (counts_need > ws.counts_n) {
            if (ws.counts) HIP_TRY(hipFree(ws.counts));
            ws.counts = nullptr;
            ws.counts_n = 0;
            HIP_TRY(hipMalloc((void **)&ws.counts, counts_need * sizeof(uint32_t)));
            ws.counts_n = counts_need;
        }
    }
    ws.n_sub = plan.n_sub;
    if (timed) HIP_TRY(hipEventRecord(ev[0], stream));

    // primary-ray tables: one entry per object, rebuilt every frame from the camera origin
    PrimScene<R> prim;
    std::memset(&prim, 0, sizeof prim);
    const SceneDev<R> &SD = scene_dev<R>(scene);
    if (cam && SD.n) {
        auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t b_sph = pad(SD.n_sph * sizeof(PrimSphereRec<R>)), b_pln = pad(SD.n_pln * sizeof(PrimPlaneRec<R>)),
                     b_cub = pad(SD.n_cub * sizeof(PrimCubeRec<R>)), b_cyl = pad(SD.n_cyl * sizeof(PrimCappedRec<R>)),
                     b_con = pad(SD.n_con * sizeof(PrimCappedRec<R>));
        const size_t b_order = pad((size_t)SD.n_grp_total * sizeof(uint32_t));
        // the spheres of the hierarchy relative to the camera (scenes with bounding-sphere tests only)
        const bool culled = (scene->sf & SF_CULL) != 0;
        const size_t b_pbnd = culled ? pad((size_t)SD.n_bnd_total * sizeof(PrimBoundRec<R>)) : 0,
                     b_pcl = culled ? pad((size_t)SD.n_cl_total * sizeof(PrimBoundRec<R>)) : 0,
                     b_pgrp = culled ? pad((size_t)SD.n_grp_total * sizeof(PrimBoundRec<R>)) : 0;
        const size_t need = b_sph + b_pln + b_cub + b_cyl + b_con + b_order + 2 * (b_pbnd + b_pcl + b_pgrp) + 256;
        if (ws.fixed) {
            if (need > ws.prim_bytes)
                return fail(RTX_ERR_INVALID, "scene too large for the fixed workspace: its primary-ray tables need %zu bytes, it holds %zu",
                            need, ws.prim_bytes);
        } else {
            const void *before = ws.prim;
            rc = grow(&ws.prim, &ws.prim_bytes, need);
            if (rc) return rc;
            if (ws.prim != before) ws.prim_scene = 0;
        }
        char *base = (char *)ws.prim;
        prim.sph = (const PrimSphereRec<R> *)base;
        prim.pln = (const PrimPlaneRec<R> *)(base + b_sph);
        prim.cub = (const PrimCubeRec<R> *)(base + b_sph + b_pln);
        prim.cyl = (const PrimCappedRec<R> *)(base + b_sph + b_pln + b_cub);
        prim.con = (const PrimCappedRec<R> *)(base + b_sph + b_pln + b_cub + b_cyl);
        char *after = base + b_sph + b_pln + b_cub + b_cyl + b_con;
        prim.grp_order = (const uint32_t *)after;
        if (culled) {
            char *at = after + b_order;
            for (ApexBounds<R> *set : { &prim.eye, &prim.light }) {
                set->bnd = (const PrimBoundRec<R> *)at;
                set->cl = (const PrimBoundRec<R> *)(at + b_pbnd);
                set->grp = (const PrimBoundRec<R> *)(at + b_pbnd + b_pcl);
                at += b_pbnd + b_pcl + b_pgrp;
            }
        }
        prim.shadow_beam_sin = R(0.1);
        prim.n_sph = SD.n_sph; prim.n_pln = SD.n_pln; prim.n_cub = SD.n_cub; prim.n_cyl = SD.n_cyl; prim.n_con = SD.n_con;
    }
    // The first launch of a frame zeroes the queue and hit counters and fills the primary-ray tables.
    // A frame with no secondary rays whose caller does not read the counters needs neither when the
    // tables already hold this scene seen from this camera origin: rendering the same view again
    // (or another row range of it) is then one launch.
    const bool tables_current = cam && SD.n && ws.prim_scene == scene->serial && ws.prim_elem == sizeof(R) &&
                                ws.prim_origin[0] == (double)cam->origin[0] && ws.prim_origin[1] == (double)cam->origin[1] &&
                                ws.prim_origin[2] == (double)cam->origin[2];
    const bool no_children = (scene->sf & (SF_REFLECTIVE | SF_TRANSPARENT)) == 0;  // nothing is ever queued
    if (!(tables_current && no_children && plan.levels == 0 && !timed)) {
        HIP_TRY(Launch<R>::frame_begin(stream, cam, SD, prim, ws.counts, (uint32_t)counts_need, (uint32_t *)ws.shards,
                                       (uint32_t)(sizeof(StatShard) * kStatShards / sizeof(uint32_t))));
        if (cam && SD.n) {
            ws.prim_scene = scene->serial;
            ws.prim_elem = sizeof(R);
            for (int k = 0; k < 3; ++k) ws.prim_origin[k] = (double)cam->origin[k];
        }
    }

    R *q_odd = (R *)ws.queues;
    R *q_even = (R *)((char *)q_odd + plan.odd_bytes);
    const SceneDev<R> &S = scene_dev<R>(scene);
    uint32_t launches = 0;
    uint64_t done = 0;
    for (uint32_t c = 0; c < n_chunks; ++c) {
        if (measured) q_odd = (R *)ws.queues;  // the levels of the last chunk may have replaced the buffer
        const uint64_t u0 = (uint64_t)c * plan.chunk_units;
        const uint64_t units = std::min<uint64_t>(plan.chunk_units, n_units - u0);
        uint32_t *counts = ws.counts + (size_t)c * counts_per_chunk;
        auto queue_at = [&](uint32_t level) {
            QueueDev<R> q;
            q.rec = (level & 1) ? q_odd : q_even;
            q.count = counts + (size_t)level * plan.n_sub * kCountStride;
            q.cap_sub = level <= plan.levels ? (uint32_t)plan.cap_sub[level] : 0u;
            q.sub_mask = plan.n_sub - 1;
            q.overflow = ctx->overflow_dev;
            queue_list_plane<R>(scene->sf, q, (uint64_t)q.cap_sub * plan.n_sub);
            return q;
        };
        const uint64_t k0 = cam ? u0 * W : u0;  // first primary ray of the chunk
        char *fb_bytes = (char *)d_fb + k0 * (format ? (size_t)(format & 0xffu) : 3 * sizeof(R));
        R *fb = (R *)fb_bytes;
        if (timed && c == 0) HIP_TRY(hipEventRecord(ev[2], stream));
        if (cam) {
            HIP_TRY(Launch<R>::primary_camera(scene->sf, stream, *cam, S, prim, map, (uint32_t)u0, (uint32_t)units, max_depth,
                                              fb_bytes, format, queue_at(1), ws.shards));
        } else {
            HIP_TRY(Launch<R>::primary_rays(scene->sf, stream, S, d_rays + k0 * 6, (uint32_t)units, max_depth, fb, queue_at(1),
                                            ws.shards));
        }
        if (timed && c == 0) HIP_TRY(hipEventRecord(ev[3], stream));
        ++launches;
        // (a measured frame lays out its second queue from the lengths of the first)
        if (plan.levels >= 1u)
            HIP_TRY(launch_list_blocks<R>(stream, scene->sf, queue_at(1),
                                          measured && max_depth >= 2 ? counts + (size_t)2 * plan.n_sub * kCountStride : nullptr));
        if (measured) {
            rc = enqueue_measured_levels<R>(ctx, scene, ws, stream, plan, counts, max_depth, fb, &launches);
            if (rc) return rc;
        } else
        for (uint32_t L = 1; L <= plan.levels; ++L) {
            // With bounding-sphere tests a block's walk is a long chain of dependent fetches whose length
            // varies widely: those launches get one workgroup per block the queue can hold (the ones past
            // its length leave at once; measured 6.9 against 7.4 ms on the 10k-object frame), the
            // others as many workgroups as the chip holds at once.
            HIP_TRY(launch_bounce<R>(scene->sf, stream, S, queue_at(L), queue_at(L + 1), max_depth - L, L, fb, ws.shards,
                                     (scene->sf & SF_LANE_TREE) ? ctx->bounce_blocks_lane
                                                                : (scene->sf & SF_CULL) ? ctx->bounce_blocks_cull : ctx->bounce_blocks,
                                     0u));
            ++launches;
            if (L < plan.levels) HIP_TRY(launch_list_blocks<R>(stream, scene->sf, queue_at(L + 1)));
        }
        done += cam ? units * W : units;
        if (progress) {
            HIP_TRY(hipStreamSynchronize(stream));
            progress(done, user);
        }
    }
    if (timed) HIP_TRY(hipEventRecord(ev[1], stream));
    if (n_chunks_out) *n_chunks_out = n_chunks;
    if (n_launch_out) *n_launch_out = launches;
    return RTX_OK;
}

// After the stream has drained: gather counters into stats; report a queue overflow.
int collect_stats(rtx_context *ctx, Slot *slot, uint32_t n_chunks, uint64_t n_primary, bool timed, rtx_stats *stats)
{
    int rc = check_overflow(ctx);
    if (rc) return rc;
    if (!stats) return RTX_OK;
    const Workspace &ws = slot->ws;
    std::vector<StatShard> shards(kStatShards);
    HIP_TRY(hipMemcpy(shards.data(), ws.shards, sizeof(StatShard) * kStatShards, hipMemcpyDeviceToHost));
    const size_t counts_per_chunk = (size_t)(kMaxLevels + 1) * ws.n_sub * kCountStride;
    std::vector<uint32_t> counts((size_t)n_chunks * counts_per_chunk);
    HIP_TRY(hipMemcpy(counts.data(), ws.counts, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::memset(stats->rays_per_level, 0, sizeof stats->rays_per_level);
    stats->rays_per_level[0] = n_primary;
    for (uint32_t c = 0; c < n_chunks; ++c)
        for (uint32_t L = 1; L < kMaxLevels; ++L)
            for (uint32_t q = 0; q < ws.n_sub; ++q)
                stats->rays_per_level[L] += counts[c * counts_per_chunk + ((size_t)L * ws.n_sub + q) * kCountStride];
    stats->color_rays = 0;
    for (uint32_t L = 0; L < kMaxLevels; ++L) stats->color_rays += stats->rays_per_level[L];
    stats->shadow_rays = 0;
    for (auto &s : shards)
        for (uint32_t L = 0; L < kMaxLevels; ++L) stats->shadow_rays += s.hits[L];
    if (timed) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, slot->ev[0], slot->ev[1]));
        stats->kernel_ms = ms;
        HIP_TRY(hipEventElapsedTime(&ms, slot->ev[2], slot->ev[3]));
        stats->primary_kernel_ms = ms;
        HIP_TRY(hipEventElapsedTime(&ms, slot->ev[3], slot->ev[1]));
        stats->levels_ms = ms;
    }
    return RTX_OK;
}

int check_camera(const rtx_camera_desc *cam, uint32_t row_begin, uint32_t row_end)
{
    if (!cam) return fail(RTX_ERR_INVALID, "camera is NULL");
    if (cam->hsize == 0 || cam->vsize == 0) return fail(RTX_ERR_INVALID, "camera has an empty canvas");
    if (row_begin > row_end || row_end > cam->vsize)
        return fail(RTX_ERR_INVALID, "rows [%u, %u) are outside the %u-row canvas", row_begin, row_end, cam->vsize);
    if ((uint64_t)cam->hsize * cam->vsize > 0x7fffffffull) return fail(RTX_ERR_INVALID, "canvas has more than 2^31 pixels");
    for (int k = 0; k < 16; ++k)
        if (!std::isfinite(cam->inverse_transform[k])) return fail(RTX_ERR_INVALID, "camera inverse_transform is not finite");
    return RTX_OK;
}

// Local rows of part `part` when the bands of `band` rows of a vsize-row image are dealt out in turn
// to n_parts parts.
uint32_t part_rows(uint32_t vsize, uint32_t band, uint32_t n_parts, uint32_t part)
{
    const uint64_t n_bands = ((uint64_t)vsize + band - 1) / band;
    if (part >= n_bands) return 0;
    const uint64_t mine = (n_bands - part + n_parts - 1) / n_parts;        // bands part, part + n_parts, ...
    const uint64_t last = part + (mine - 1) * n_parts;                      // the last of them
    const uint64_t last_rows = std::min<uint64_t>(band, vsize - last * band);
    return (uint32_t)((mine - 1) * band + last_rows);
}

bool valid_format8(uint32_t format) { return format == RTX_PIXELS_RGB8 || format == RTX_PIXELS_RGBA8 || format == RTX_PIXELS_PPM8; }

// The rows one call covers: a plain range [row_begin, row_end) or the bands of one part.
struct Rows {
    RowMap map;
    uint32_t n_local;
};
Rows row_range(uint32_t row_begin, uint32_t row_end) { return { { row_begin, 1u, 1u, 0u }, row_end - row_begin }; }

template <typename R>
int render_impl(rtx_context *ctx, const rtx_scene *scene, const rtx_camera_desc *cam, uint32_t max_depth, const Rows &rows,
                void *host_out, void *dev_out, hipStream_t user_stream, rtx_progress_fn progress, void *user, rtx_stats *stats,
                uint32_t channels8 = 0)
{
    // channels8 != 0: an RTX_PIXELS_* format; the output (host or device) receives 8-bit pixels instead of R values
    const auto t0 = std::chrono::steady_clock::now();
    int rc = check_overflow(ctx);
    if (rc) return rc;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    SlotLease lease(ctx);
    rc = slot_acquire(ctx, &lease.slot);
    if (rc) return rc;
    Slot *slot = lease.slot;
    const uint64_t W = cam->hsize;
    const uint64_t n_pix = (uint64_t)rows.n_local * W;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n_pix == 0) return RTX_OK;
    hipStream_t stream = user_stream ? user_stream : slot->stream;
    rc = slot_order_after_pending(slot, stream);
    if (rc) return rc;
    // An 8-bit frame whose pixels get nothing from deeper levels is converted by the primary kernel
    // as it stores (no R-valued frame exists at all); otherwise the levels accumulate in an R-valued
    // frame and one more pass converts it.
    const bool direct8 = channels8 != 0 && (scene->sf & (SF_REFLECTIVE | SF_TRANSPARENT)) == 0;
    const size_t bytes8 = (size_t)(n_pix * (channels8 & 0xffu));
    void *d_pixels8 = nullptr;  // where the 8-bit image goes on the device
    if (channels8) {
        d_pixels8 = dev_out;
        if (!d_pixels8) {
            rc = grow(&slot->aux, &slot->aux_bytes, bytes8);
            if (rc) return rc;
            d_pixels8 = slot->aux;
        }
    }
    R *d_fb = channels8 ? nullptr : (R *)dev_out;
    if (!d_fb && !direct8) {
        rc = grow(&slot->fb, &slot->fb_bytes, (size_t)(n_pix * 3 * sizeof(R)));
        if (rc) return rc;
        d_fb = (R *)slot->fb;
    }
    const CameraDev<R> cd = make_camera<R>(cam);
    uint32_t n_chunks = 0, n_launch = 0;
    // without a stream of the caller's there is nothing the caller could wait on: the call waits itself
    const bool sync = host_out != nullptr || stats != nullptr || user_stream == nullptr;
    const bool timed = stats != nullptr;
    rc = enqueue_frame<R>(ctx, scene, slot->ws, slot->ev, stream, &cd, nullptr, rows.map, rows.n_local, max_depth,
                          direct8 ? d_pixels8 : (void *)d_fb, direct8 ? channels8 : 0u, progress, user, timed, &n_chunks, &n_launch, sync);
    if (rc) return rc;
    if (channels8 && !direct8) {
        HIP_TRY(Launch<R>::pack8(stream, d_fb, (uint32_t)n_pix, channels8, (uint8_t *)d_pixels8));
        ++n_launch;
    }
    if (!sync) {
        HIP_TRY(hipEventRecord(slot->done, stream));
        slot->pending = true;
        slot_settled(slot);
        return RTX_OK;
    }
    if (host_out && channels8) {
        rc = copy_to_host(slot, stream, host_out, d_pixels8, bytes8);
        if (rc) return rc;
    } else if (host_out) {
        rc = copy_to_host(slot, stream, host_out, d_fb, (size_t)(n_pix * 3 * sizeof(R)));
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    slot_settled(slot);
    rc = collect_stats(ctx, slot, n_chunks, n_pix, timed, stats);
    if (rc) return rc;
    if (stats) {
        stats->n_chunks = n_chunks;
        stats->n_launches = n_launch;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RTX_OK;
}

template <typename R>
int trace_impl(rtx_context *ctx, const rtx_scene *scene, const double *rays, size_t n, uint32_t max_depth,
               double *out_rgb, rtx_stats *stats)
{
    const auto t0 = std::chrono::steady_clock::now();
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    SlotLease lease(ctx);
    int rc = slot_acquire(ctx, &lease.slot);
    if (rc) return rc;
    Slot *slot = lease.slot;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n == 0) return RTX_OK;
    const size_t ray_bytes = n * 6 * sizeof(double);
    rc = grow(&slot->aux, &slot->aux_bytes, ray_bytes);
    if (rc) return rc;
    rc = grow(&slot->fb, &slot->fb_bytes, n * 3 * sizeof(R));
    if (rc) return rc;
    hipStream_t stream = slot->stream;
    rc = slot_order_after_pending(slot, stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(slot->aux, rays, ray_bytes, hipMemcpyHostToDevice, stream));
    uint32_t n_chunks = 0, n_launch = 0;
    rc = enqueue_frame<R>(ctx, scene, slot->ws, slot->ev, stream, nullptr, (const double *)slot->aux, RowMap{ 0u, 1u, 1u, 0u }, n, max_depth,
                          slot->fb, 0u, nullptr, nullptr, true, &n_chunks, &n_launch, true);
    if (rc) return rc;
    std::vector<R> tmp;
    void *dst = out_rgb;
    if (sizeof(R) != sizeof(double)) {
        tmp.resize(n * 3);
        dst = tmp.data();
    }
    HIP_TRY(hipMemcpyAsync(dst, slot->fb, n * 3 * sizeof(R), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    slot_settled(slot);
    if (sizeof(R) != sizeof(double))
        for (size_t i = 0; i < n * 3; ++i) out_rgb[i] = (double)tmp[i];
    rc = collect_stats(ctx, slot, n_chunks, n, true, stats);
    if (rc) return rc;
    if (stats) {
        stats->n_chunks = n_chunks;
        stats->n_launches = n_launch;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RTX_OK;
}

// Layout of the file-scope device array for one frame of rtx_render_device_noalloc: hit counters,
// queue length counters, primary-ray tables, then both ray queues.
constexpr size_t kNoAllocCountBytes = (size_t)32 << 20;
constexpr size_t kNoAllocPrimBytes = (size_t)8 << 20;

template <typename R>
int render_noalloc(rtx_context *ctx, const rtx_scene *scene, const rtx_camera_desc *cam, uint32_t max_depth, uint32_t row_begin,
                          uint32_t row_end, void *d_out, hipStream_t stream)
{
    Workspace ws;
    char *base = (char *)ctx->scratch;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base + off;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    ws.fixed = true;
    ws.shards = (StatShard *)take(sizeof(StatShard) * kStatShards);
    ws.counts = (uint32_t *)take(kNoAllocCountBytes);
    ws.counts_n = kNoAllocCountBytes / sizeof(uint32_t);
    ws.prim = take(kNoAllocPrimBytes);
    ws.prim_bytes = kNoAllocPrimBytes;
    ws.queues = base + off;
    ws.queue_bytes = ctx->scratch_bytes - off;
    const CameraDev<R> cd = make_camera<R>(cam);
    return enqueue_frame<R>(ctx, scene, ws, nullptr, stream, &cd, nullptr, RowMap{ row_begin, 1u, 1u, 0u }, row_end - row_begin, max_depth,
                            d_out, 0u, nullptr, nullptr, false, nullptr, nullptr);
}

}  // namespace

extern "C" {

const char *rtx_last_error(void) { return tl_error.c_str(); }

int rtx_context_create(int device_id, rtx_context **out)
{
    if (!out) return fail(RTX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(RTX_ERR_DEVICE, "no HIP device is visible: the render loop has no CPU fallback");
    if (device_id < 0) HIP_TRY(hipGetDevice(&device_id));
    if (device_id >= count) return fail(RTX_ERR_INVALID, "device %d does not exist (%d visible)", device_id, count);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RTX_ERR_DEVICE, "device %d is %s; the kernels are built for gfx950 only", device_id, prop.gcnArchName);
    std::unique_ptr<rtx_context> ctx(new rtx_context);
    ctx->device = device_id;
    ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->clock_mhz = prop.clockRate / 1000;
    ctx->name = prop.name;
    // Queues are sized for the worst case (every hit reflects and refracts down to the depth
    // limit): a 4K depth-5 frame of a scene with glass needs about 26 GB.  A sixth of the device
    // memory (48 GB of 288) lets such a frame run as one chunk and leaves room for several
    // concurrent render calls.
    ctx->workspace_limit = std::max<size_t>((size_t)8 << 30, prop.totalGlobalMem / 6);
    // Workgroups of a bounce launch: each loops over its share of a sub-queue's rays.  The chip holds about
    // 10 of them per CU at once (f64: 2 waves of 85 registers each); three times that many, shorter ones
    // keep it filled to the end of the launch: pieces of 128 rays differ several-fold in what they cost
    // (4K glass frame: 2.04 ms with 8 per CU, 1.93 with 16, 1.84 with 30, 1.81-1.88 with 50-100; one per
    // piece, 2.57).  Handing the pieces out by ticket instead was slower than this fixed split.
    ctx->bounce_blocks = (uint32_t)ctx->cu_count * 30u;
    ctx->bounce_blocks_lane = (uint32_t)ctx->cu_count;
    {
        DeviceGuard guard(device_id);
        if (!guard.ok) return fail(RTX_ERR_DEVICE, "hipSetDevice(%d) failed", device_id);
        // this device's copy of the file-scope array that rtx_render_device_noalloc works in
        HIP_TRY(scratch_address(&ctx->scratch, &ctx->scratch_bytes));
        HIP_TRY(hipHostMalloc((void **)&ctx->overflow_host, 256, hipHostMallocMapped));
        *ctx->overflow_host = 0u;
        const hipError_t e = hipHostGetDevicePointer((void **)&ctx->overflow_dev, ctx->overflow_host, 0);
        if (e != hipSuccess) {
            (void)hipHostFree(ctx->overflow_host);
            return fail(RTX_ERR_DEVICE, "hipHostGetDevicePointer failed: %s", hipGetErrorString(e));
        }
    }
    *out = ctx.release();
    return RTX_OK;
}

void rtx_context_destroy(rtx_context *ctx)
{
    if (!ctx) return;
    {
        DeviceGuard guard(ctx->device);
        for (auto &s : ctx->slots) s->release();
        if (ctx->overflow_host) (void)hipHostFree(ctx->overflow_host);
    }
    delete ctx;
}

int rtx_context_synchronize(rtx_context *ctx)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "ctx is NULL");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    std::vector<hipEvent_t> waits;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (auto &s : ctx->slots)
            if (s->pending) waits.push_back(s->done);
    }
    for (hipEvent_t e : waits) HIP_TRY(hipEventSynchronize(e));
    {   // what was waited for is no longer in flight: those slots' queues may be given back when memory runs short
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (auto &s : ctx->slots)
            if (!s->busy && s->pending && hipEventQuery(s->done) == hipSuccess) s->pending = false;
    }
    return check_overflow(ctx);
}

int rtx_context_set_debug_queue_divisor(rtx_context *ctx, uint32_t divisor)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "ctx is NULL");
    if (divisor == 0) return fail(RTX_ERR_INVALID, "divisor must be at least 1");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->debug_queue_divisor = divisor;
    return RTX_OK;
}

int rtx_context_set_measured_queues(rtx_context *ctx, size_t worst_case_bytes_above)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->measured_above = worst_case_bytes_above;
    return RTX_OK;
}

int rtx_context_queue_bytes(rtx_context *ctx, size_t *bytes)
{
    if (!ctx || !bytes) return fail(RTX_ERR_INVALID, "ctx or bytes is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    size_t total = 0;
    for (auto &s : ctx->slots) total += s->ws.queue_bytes + s->ws.queue_even_bytes;
    *bytes = total;
    return RTX_OK;
}

int rtx_context_set_workspace_limit(rtx_context *ctx, size_t bytes)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "ctx is NULL");
    if (bytes < ((size_t)1 << 20)) return fail(RTX_ERR_INVALID, "workspace limit below 1 MiB");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->workspace_limit = bytes;
    return RTX_OK;
}

int rtx_context_set_lane_walk(rtx_context *ctx, int enabled)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->lane_tree = enabled != 0;
    return RTX_OK;
}

int rtx_context_set_cull_threshold(rtx_context *ctx, uint32_t min_bounded_objects)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->cull_min_objects = min_bounded_objects;
    return RTX_OK;
}

int rtx_scene_upload(rtx_context *ctx, const rtx_scene_desc *d, rtx_scene **out)
{
    if (!ctx || !d || !out) return fail(RTX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    const uint32_t n = d->n_objects;
    if (n > 0) {
        if (!d->shape_type || !d->inverse || !d->inverse_transposed || !d->ymin || !d->ymax || !d->closed || !d->color ||
            !d->ambient || !d->diffuse || !d->specular || !d->shininess || !d->reflective || !d->transparency ||
            !d->refractive_index || !d->casts_shadow || !d->pattern_type || !d->pattern_inverse || !d->pattern_color1 ||
            !d->pattern_color2)
            return fail(RTX_ERR_INVALID, "scene description has a NULL array");
        // World::shade_hit and is_shadowed index lights[0] (src/scenes/world.rs:123,147): the
        // reference panics on the first hit of a world without lights.
        if (d->n_lights == 0 || !d->light_position || !d->light_intensity)
            return fail(RTX_ERR_INVALID, "world has objects but no light: the reference panics at lights[0]");
    }
    for (uint32_t i = 0; i < n; ++i) {
        if (d->shape_type[i] > RTX_CONE) return fail(RTX_ERR_INVALID, "object %u: unknown shape tag %u", i, d->shape_type[i]);
        if (d->pattern_type[i] > RTX_TEST) return fail(RTX_ERR_INVALID, "object %u: unknown pattern tag %u", i, d->pattern_type[i]);
        for (int k = 0; k < 16; ++k)
            if (!std::isfinite(d->inverse[(size_t)i * 16 + k]) || !std::isfinite(d->inverse_transposed[(size_t)i * 16 + k]))
                return fail(RTX_ERR_INVALID, "object %u: transform is not finite", i);
        if (d->chain_depth && d->chain_depth[i]) {
            if (d->chain_depth[i] > kMaxChain)
                return fail(RTX_ERR_INVALID, "object %u sits %u groups deep; at most %u levels are supported", i, d->chain_depth[i],
                            kMaxChain);
            if (!d->chain_first || !d->chain_inverse) return fail(RTX_ERR_INVALID, "chain_depth without chain_first / chain_inverse");
            for (uint32_t k = 0; k < 16u * (d->chain_depth[i] + 1u); ++k)
                if (!std::isfinite(d->chain_inverse[(size_t)d->chain_first[i] * 16 + k]))
                    return fail(RTX_ERR_INVALID, "object %u: a group transform is not finite", i);
        }
    }
    // Refraction containers (src/scenes/world.rs:253-285) find and remove objects by value
    // equality, so k objects that compare equal toggle one list entry k times per shared t:
    // an even class never appears in the list, an odd one appears once.  Its lowest index is
    // the only member that can be the hit (ties go to the lowest index), so it carries the flag.
    std::vector<uint8_t> container(n, 1);
    if (d->eq_class) {
        std::map<uint32_t, std::pair<uint32_t, uint32_t>> classes;  // class -> (first index, members)
        for (uint32_t i = 0; i < n; ++i) {
            auto it = classes.find(d->eq_class[i]);
            if (it == classes.end()) classes.emplace(d->eq_class[i], std::make_pair(i, 1u));
            else it->second.second++;
        }
        for (uint32_t i = 0; i < n; ++i) {
            const auto &c = classes[d->eq_class[i]];
            container[i] = (c.first == i && (c.second & 1u)) ? 1 : 0;
        }
    }
    uint32_t sf = n ? SF_ALL_CAST : 0u;
    for (uint32_t i = 0; i < n; ++i) {
        if (d->reflective[i] != 0.0) sf |= SF_REFLECTIVE;
        if (d->transparency[i] != 0.0) sf |= SF_TRANSPARENT;
        if (!d->casts_shadow[i]) sf &= ~SF_ALL_CAST;
        if (d->pattern_type[i] != RTX_SOLID) sf |= SF_PATTERN;
        if (d->chain_depth && d->chain_depth[i]) sf |= SF_CHAIN;
    }
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    std::unique_ptr<rtx_scene> s(new rtx_scene);
    static std::atomic<uint64_t> next_serial{ 1 };
    s->serial = next_serial.fetch_add(1);
    s->ctx = ctx;
    s->n = n;
    // Bounding-sphere tests cost about 20 instructions per object and ray: they pay once most
    // rays miss most objects, which a handful of large objects in view does not give.
    uint32_t n_bounded = 0;
    const std::vector<Bound> bounds = scene_bounds(d, &n_bounded);
    if (n_bounded >= ctx->cull_min_objects) sf |= SF_CULL;
    Tables<double> t64;
    Tables<float> t32;
    fill_tables<double>(d, container, bounds, (sf & SF_CULL) != 0, t64);
    fill_tables<float>(d, container, bounds, (sf & SF_CULL) != 0, t32);
    // secondary rays of a scene whose packed hierarchy fits a workgroup's LDS are walked lane by lane
    QuantTree qt;
    if ((sf & SF_CULL) && ctx->lane_tree) qt = quantize_tree(t64);
    if (!qt.nodes.empty()) sf |= SF_LANE_TREE;
    s->sf = sf;
    int rc = upload_tables<double>(t64, qt, sf, d, &s->blob64, &s->dev64);
    if (!rc) rc = upload_tables<float>(t32, qt, sf, d, &s->blob32, &s->dev32);
    if (rc) {
        rtx_scene_destroy(s.release());
        return rc;
    }
    *out = s.release();
    return RTX_OK;
}

int rtx_scene_features(const rtx_scene *scene, uint32_t *bits)
{
    if (!scene || !bits) return fail(RTX_ERR_INVALID, "scene or bits is NULL");
    *bits = scene->sf;
    return RTX_OK;
}

void rtx_scene_destroy(rtx_scene *s)
{
    if (!s) return;
    {
        DeviceGuard guard(s->ctx->device);
        if (s->blob64) (void)hipFree(s->blob64);
        if (s->blob32) (void)hipFree(s->blob32);
    }
    delete s;
}

int rtx_render(rtx_context *ctx, const rtx_scene *scene, const rtx_camera_desc *camera, uint32_t max_depth, int precision,
               uint32_t row_begin, uint32_t row_end, void *out_rgb, size_t out_bytes, rtx_progress_fn progress, void *user,
               rtx_stats *stats)
{
    if (!ctx || !scene || scene->ctx != ctx) return fail(RTX_ERR_INVALID, "scene does not belong to this context");
    int rc = check_camera(camera, row_begin, row_end);
    if (rc) return rc;
    if (precision != RTX_F64 && precision != RTX_F32) return fail(RTX_ERR_INVALID, "unknown precision %d", precision);
    const size_t elem = precision == RTX_F64 ? sizeof(double) : sizeof(float);
    const size_t need = (size_t)(row_end - row_begin) * camera->hsize * 3 * elem;
    if (need && (!out_rgb || out_bytes < need)) return fail(RTX_ERR_INVALID, "output buffer holds %zu bytes, %zu needed", out_bytes, need);
    if (precision == RTX_F64)
        return render_impl<double>(ctx, scene, camera, max_depth, row_range(row_begin, row_end), out_rgb, nullptr, nullptr, progress, user, stats);
    return render_impl<float>(ctx, scene, camera, max_depth, row_range(row_begin, row_end), out_rgb, nullptr, nullptr, progress, user, stats);
}

int rtx_render_8bit(rtx_context *ctx, const rtx_scene *scene, const rtx_camera_desc *camera, uint32_t max_depth, int precision,
                    uint32_t row_begin, uint32_t row_end, uint32_t format, uint8_t *out, size_t out_bytes,
                    rtx_progress_fn progress, void *user, rtx_stats *stats)
{
    if (!ctx || !scene || scene->ctx != ctx) return fail(RTX_ERR_INVALID, "scene does not belong to this context");
    int rc = check_camera(camera, row_begin, row_end);
    if (rc) return rc;
    if (precision != RTX_F64 && precision != RTX_F32) return fail(RTX_ERR_INVALID, "unknown precision %d", precision);
    if (!valid_format8(format))
        return fail(RTX_ERR_INVALID, "format must be RTX_PIXELS_RGB8 (3), RTX_PIXELS_RGBA8 (4) or RTX_PIXELS_PPM8 (0x103), not %u", format);
    const size_t need = (size_t)(row_end - row_begin) * camera->hsize * (format & 0xffu);
    if (need && (!out || out_bytes < need)) return fail(RTX_ERR_INVALID, "output buffer holds %zu bytes, %zu needed", out_bytes, need);
    if (need == 0) {
        if (stats) std::memset(stats, 0, sizeof *stats);
        return RTX_OK;
    }
    if (precision == RTX_F64)
        return render_impl<double>(ctx, scene, camera, max_depth, row_range(row_begin, row_end), out, nullptr, nullptr, progress, user,
                                   stats, format);
    return render_impl<float>(ctx, scene, camera, max_depth, row_range(row_begin, row_end), out, nullptr, nullptr, progress, user, stats,
                              format);
}

int rtx_render_device(rtx_context *ctx, const rtx_scene *scene, const rtx_camera_desc *camera, uint32_t max_depth,
                      int precision, uint32_t row_begin, uint32_t row_end, void *d_out_rgb, size_t out_bytes, void *stream,
                      rtx_stats *stats)
{
    if (!ctx || !scene || scene->ctx != ctx) return fail(RTX_ERR_INVALID, "scene does not belong to this context");
    int rc = check_camera(camera, row_begin, row_end);
    if (rc) return rc;
    if (precision != RTX_F64 && precision != RTX_F32) return fail(RTX_ERR_INVALID, "unknown precision %d", precision);
    const size_t elem = precision == RTX_F64 ? sizeof(double) : sizeof(float);
    const size_t need = (size_t)(row_end - row_begin) * camera->hsize * 3 * elem;
    if (need && (!d_out_rgb || out_bytes < need)) return fail(RTX_ERR_INVALID, "output buffer holds %zu bytes, %zu needed", out_bytes, need);
    if (precision == RTX_F64)
        return render_impl<double>(ctx, scene, camera, max_depth, row_range(row_begin, row_end), nullptr, d_out_rgb, (hipStream_t)stream,
                                   nullptr, nullptr, stats);
    return render_impl<float>(ctx, scene, camera, max_depth, row_range(row_begin, row_end), nullptr, d_out_rgb, (hipStream_t)stream, nullptr,
                              nullptr, stats);
}

uint32_t rtx_part_rows(uint32_t vsize, uint32_t band_rows, uint32_t n_parts, uint32_t part)
{
    if (band_rows == 0 || n_parts == 0 || part >= n_parts) return 0;
    return part_rows(vsize, band_rows, n_parts, part);
}

int rtx_render_part_device(rtx_context *ctx, const rtx_scene *scene, const rtx_camera_desc *camera, uint32_t max_depth,
                           int precision, uint32_t band_rows, uint32_t n_parts, uint32_t part, uint32_t format, void *d_out,
                           size_t out_bytes, void *stream, rtx_stats *stats)
{
    if (!ctx || !scene || scene->ctx != ctx) return fail(RTX_ERR_INVALID, "scene does not belong to this context");
    int rc = check_camera(camera, 0, camera ? camera->vsize : 0);
    if (rc) return rc;
    if (precision != RTX_F64 && precision != RTX_F32) return fail(RTX_ERR_INVALID, "unknown precision %d", precision);
    if (band_rows == 0 || n_parts == 0 || part >= n_parts)
        return fail(RTX_ERR_INVALID, "part %u of %u parts with bands of %u rows is not a part of the frame", part, n_parts, band_rows);
    if (format != 0 && !valid_format8(format))
        return fail(RTX_ERR_INVALID, "format must be 0 (values), RTX_PIXELS_RGB8 (3), RTX_PIXELS_RGBA8 (4) or RTX_PIXELS_PPM8 (0x103), not %u", format);
    const Rows rows = { { 0u, band_rows, n_parts, part }, part_rows(camera->vsize, band_rows, n_parts, part) };
    const size_t elem = precision == RTX_F64 ? sizeof(double) : sizeof(float);
    const size_t need = (size_t)rows.n_local * camera->hsize * (format ? (size_t)(format & 0xffu) : 3 * elem);
    if (need && (!d_out || out_bytes < need)) return fail(RTX_ERR_INVALID, "output buffer holds %zu bytes, %zu needed", out_bytes, need);
    if (need == 0) {
        if (stats) std::memset(stats, 0, sizeof *stats);
        return RTX_OK;
    }
    if (precision == RTX_F64)
        return render_impl<double>(ctx, scene, camera, max_depth, rows, nullptr, d_out, (hipStream_t)stream, nullptr, nullptr, stats, format);
    return render_impl<float>(ctx, scene, camera, max_depth, rows, nullptr, d_out, (hipStream_t)stream, nullptr, nullptr, stats, format);
}

int rtx_render_device_noalloc(rtx_context *ctx, const rtx_scene *scene, const rtx_camera_desc *camera, uint32_t max_depth,
                              int precision, uint32_t row_begin, uint32_t row_end, void *d_out_rgb, size_t out_bytes, void *stream)
{
    if (!ctx || !scene || scene->ctx != ctx) return fail(RTX_ERR_INVALID, "scene does not belong to this context");
    int rc = check_camera(camera, row_begin, row_end);
    if (rc) return rc;
    if (precision != RTX_F64 && precision != RTX_F32) return fail(RTX_ERR_INVALID, "unknown precision %d", precision);
    const size_t elem = precision == RTX_F64 ? sizeof(double) : sizeof(float);
    const size_t need = (size_t)(row_end - row_begin) * camera->hsize * 3 * elem;
    if (need && (!d_out_rgb || out_bytes < need)) return fail(RTX_ERR_INVALID, "output buffer holds %zu bytes, %zu needed", out_bytes, need);
    if (!ctx->scratch || ctx->scratch_bytes < kNoAllocCountBytes + kNoAllocPrimBytes + ((size_t)1 << 20))
        return fail(RTX_ERR_DEVICE, "the context has no fixed workspace");
    rc = check_overflow(ctx);
    if (rc) return rc;
    if (need == 0) return RTX_OK;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    if (precision == RTX_F64)
        return render_noalloc<double>(ctx, scene, camera, max_depth, row_begin, row_end, d_out_rgb, (hipStream_t)stream);
    return render_noalloc<float>(ctx, scene, camera, max_depth, row_begin, row_end, d_out_rgb, (hipStream_t)stream);
}

int rtx_trace_rays(rtx_context *ctx, const rtx_scene *scene, const double *rays, size_t n, uint32_t max_depth, int precision,
                   double *out_rgb, rtx_stats *stats)
{
    if (!ctx || !scene || scene->ctx != ctx) return fail(RTX_ERR_INVALID, "scene does not belong to this context");
    if (n && (!rays || !out_rgb)) return fail(RTX_ERR_INVALID, "NULL ray or output array");
    if (n > 0x7fffff00ull) return fail(RTX_ERR_INVALID, "too many rays in one batch");
    if (precision == RTX_F64) return trace_impl<double>(ctx, scene, rays, n, max_depth, out_rgb, stats);
    if (precision == RTX_F32) return trace_impl<float>(ctx, scene, rays, n, max_depth, out_rgb, stats);
    return fail(RTX_ERR_INVALID, "unknown precision %d", precision);
}

int rtx_probe_rays(rtx_context *ctx, const rtx_scene *scene, const double *rays, size_t n, rtx_probe *out)
{
    if (!ctx || !scene || scene->ctx != ctx) return fail(RTX_ERR_INVALID, "scene does not belong to this context");
    if (n == 0) return RTX_OK;
    if (!rays || !out) return fail(RTX_ERR_INVALID, "NULL ray or output array");
    if (n > 0x7fffff00ull) return fail(RTX_ERR_INVALID, "too many rays in one batch");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    SlotLease lease(ctx);
    int rc = slot_acquire(ctx, &lease.slot);
    if (rc) return rc;
    Slot *slot = lease.slot;
    const size_t ray_bytes = n * 6 * sizeof(double);
    const size_t ray_pad = (ray_bytes + 255) & ~(size_t)255;
    rc = grow(&slot->aux, &slot->aux_bytes, ray_pad + n * sizeof(rtx_probe));
    if (rc) return rc;
    rc = slot_order_after_pending(slot, slot->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(slot->aux, rays, ray_bytes, hipMemcpyHostToDevice, slot->stream));
    rtx_probe *d_out = (rtx_probe *)((char *)slot->aux + ray_pad);
    HIP_TRY(launch_probe(scene->sf, grid_for(ctx, n), slot->stream, scene->dev64, (const double *)slot->aux, (uint32_t)n, d_out));
    HIP_TRY(hipMemcpyAsync(out, d_out, n * sizeof(rtx_probe), hipMemcpyDeviceToHost, slot->stream));
    HIP_TRY(hipStreamSynchronize(slot->stream));
    slot_settled(slot);
    return RTX_OK;
}

int rtx_selfcheck_division(rtx_context *ctx, uint64_t seed, uint32_t n, int narrow, uint64_t *mismatches)
{
    if (!ctx || !mismatches) return fail(RTX_ERR_INVALID, "ctx or mismatches is NULL");
    if (n == 0 || n > 0x7fffff00u) return fail(RTX_ERR_INVALID, "n must be between 1 and 2^31 - 256");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    SlotLease lease(ctx);
    int rc = slot_acquire(ctx, &lease.slot);
    if (rc) return rc;
    Slot *slot = lease.slot;
    rc = grow(&slot->aux, &slot->aux_bytes, 256);
    if (rc) return rc;
    rc = slot_order_after_pending(slot, slot->stream);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(slot->aux, 0, sizeof(unsigned long long), slot->stream));
    HIP_TRY(launch_check_div3(slot->stream, seed, n, narrow ? 1u : 0u, (unsigned long long *)slot->aux));
    unsigned long long count = 0;
    HIP_TRY(hipMemcpyAsync(&count, slot->aux, sizeof count, hipMemcpyDeviceToHost, slot->stream));
    HIP_TRY(hipStreamSynchronize(slot->stream));
    slot_settled(slot);
    *mismatches = count;
    return RTX_OK;
}

int rtx_selfcheck_pow(rtx_context *ctx, uint64_t seed, uint32_t n, double *max_abs_diff, double *max_rel_diff)
{
    if (!ctx || !max_abs_diff || !max_rel_diff) return fail(RTX_ERR_INVALID, "ctx or an output pointer is NULL");
    if (n == 0 || n > 0x7fffff00u) return fail(RTX_ERR_INVALID, "n must be between 1 and 2^31 - 256");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    SlotLease lease(ctx);
    int rc = slot_acquire(ctx, &lease.slot);
    if (rc) return rc;
    Slot *slot = lease.slot;
    rc = grow(&slot->aux, &slot->aux_bytes, 256);
    if (rc) return rc;
    rc = slot_order_after_pending(slot, slot->stream);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(slot->aux, 0, 2 * sizeof(unsigned long long), slot->stream));
    HIP_TRY(launch_check_pow(slot->stream, seed, n, (unsigned long long *)slot->aux));
    double out[2] = { 0.0, 0.0 };
    HIP_TRY(hipMemcpyAsync(out, slot->aux, sizeof out, hipMemcpyDeviceToHost, slot->stream));
    HIP_TRY(hipStreamSynchronize(slot->stream));
    slot_settled(slot);
    *max_abs_diff = out[0];
    *max_rel_diff = out[1];
    return RTX_OK;
}

int rtx_generate_rays(rtx_context *ctx, const rtx_camera_desc *camera, uint64_t first_pixel, size_t n, double *out)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "ctx is NULL");
    int rc = check_camera(camera, 0, camera ? camera->vsize : 0);
    if (rc) return rc;
    if (n == 0) return RTX_OK;
    if (!out) return fail(RTX_ERR_INVALID, "out is NULL");
    if (first_pixel + n > (uint64_t)camera->hsize * camera->vsize) return fail(RTX_ERR_INVALID, "pixel range is outside the canvas");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    SlotLease lease(ctx);
    rc = slot_acquire(ctx, &lease.slot);
    if (rc) return rc;
    Slot *slot = lease.slot;
    rc = grow(&slot->aux, &slot->aux_bytes, n * 6 * sizeof(double));
    if (rc) return rc;
    rc = slot_order_after_pending(slot, slot->stream);
    if (rc) return rc;
    const CameraDev<double> cd = make_camera<double>(camera);
    HIP_TRY(Launch<double>::raygen(grid_for(ctx, n), slot->stream, cd, (uint32_t)first_pixel, (uint32_t)n, (double *)slot->aux));
    HIP_TRY(hipMemcpyAsync(out, slot->aux, n * 6 * sizeof(double), hipMemcpyDeviceToHost, slot->stream));
    HIP_TRY(hipStreamSynchronize(slot->stream));
    slot_settled(slot);
    return RTX_OK;
}

int rtx_quantize_u8_device(rtx_context *ctx, const void *d_rgb, int precision, size_t n_values, uint8_t *d_out, void *stream)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "ctx is NULL");
    if (n_values == 0) return RTX_OK;
    if (!d_rgb || !d_out) return fail(RTX_ERR_INVALID, "NULL device pointer");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    hipStream_t s = (hipStream_t)stream;
    SlotLease lease(ctx);
    if (!s) {
        int rc = slot_acquire(ctx, &lease.slot);
        if (rc) return rc;
        s = lease.slot->stream;
    }
    const uint32_t grid = grid_for(ctx, n_values);
    if (precision == RTX_F64) HIP_TRY(Launch<double>::quantize(grid, s, (const double *)d_rgb, n_values, d_out));
    else if (precision == RTX_F32) HIP_TRY(Launch<float>::quantize(grid, s, (const float *)d_rgb, n_values, d_out));
    else return fail(RTX_ERR_INVALID, "unknown precision %d", precision);
    if (!stream) HIP_TRY(hipStreamSynchronize(s));
    return RTX_OK;
}

int rtx_device_info(rtx_context *ctx, char *name, size_t name_bytes, int *compute_units, int *clock_mhz)
{
    if (!ctx) return fail(RTX_ERR_INVALID, "ctx is NULL");
    if (name && name_bytes) snprintf(name, name_bytes, "%s", ctx->name.c_str());
    if (compute_units) *compute_units = ctx->cu_count;
    if (clock_mhz) *clock_mhz = ctx->clock_mhz;
    return RTX_OK;
}

}  // extern "C"
