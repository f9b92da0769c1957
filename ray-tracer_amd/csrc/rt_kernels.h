// rt_kernels.h -- the render kernels and their launchers, templated on the scalar type.
//
// Kernel map (the reference's recursion, src/scenes/world.rs:121-190, unrolled by level):
//   k_primary  one pixel per lane: ray_for_pixel -> closest hit over the object table ->
//              shadow ray -> Phong; stores the pixel's own lighting term and appends the
//              reflection / refraction rays it spawns to the level-1 queue.
//   k_bounce   one queued ray per lane, same work; adds weight * lighting to its pixel and
//              appends to the next level's queue.  Launched once per recursion level.
// Queue appends are aggregated per wave: one atomic per 64-lane wave reserves a contiguous
// run of slots, each lane writes its slot with 16-byte stores.
#pragma once

#include "rt_device.h"
#include "rt_launch.h"

namespace rtx {

template <typename R> struct Vec2T;
template <> struct Vec2T<double> { using type = double2; };
template <> struct Vec2T<float> { using type = float2; };

RT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <typename R> RT_DEV uint32_t pixel_bits_load(R v);
template <> RT_DEV uint32_t pixel_bits_load<double>(double v) { return (uint32_t)__double_as_longlong(v); }
template <> RT_DEV uint32_t pixel_bits_load<float>(float v) { return __float_as_uint(v); }
template <typename R> RT_DEV R pixel_bits_store(uint32_t p);
template <> RT_DEV double pixel_bits_store<double>(uint32_t p) { return __longlong_as_double((long long)p); }
template <> RT_DEV float pixel_bits_store<float>(uint32_t p) { return __uint_as_float(p); }

// Append (o, d, weight, pixel) to sub-queue `sub` for every lane with `active` set.  Must be
// reached by all lanes of the wave (inactive ones pass active = false): the wave reserves its
// slots with one atomic.
template <typename R>
RT_DEV void queue_push(const QueueDev<R> &q, uint32_t sub, bool active, V3<R> o, V3<R> d, R weight, uint32_t pixel,
                       StatShard *__restrict__ stats)
{
    const unsigned long long mask = __ballot(active);
    if (mask == 0ull) return;
    const uint32_t lane = lane_id();
    const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(q.count + (size_t)sub * kCountStride, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    if (active) {
        const uint32_t idx = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (idx < q.cap_sub) {
            using V2 = typename Vec2T<R>::type;
            V2 *dst = reinterpret_cast<V2 *>(q.rec) + ((size_t)sub * q.cap_sub + idx) * 4;
            V2 a, b, c, e;
#ifdef RTX_EXPERIMENT_ROUND_RAYS
            o = { R(float(o.x)), R(float(o.y)), R(float(o.z)) };
            d = { R(float(d.x)), R(float(d.y)), R(float(d.z)) };
#endif
            a.x = o.x; a.y = o.y;
            b.x = o.z; b.y = d.x;
            c.x = d.y; c.y = d.z;
            e.x = weight; e.y = pixel_bits_store<R>(pixel);
            dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = e;
        } else {
            atomicAdd(&stats->overflow, 1ull);  // cannot happen: queues are sized for the worst case
        }
    }
}

// Count the shaded hits of this wave: one atomic per wave that has any, spread over shards.
RT_DEV void count_hits(StatShard *__restrict__ stats, uint32_t group, uint32_t level, bool hit)
{
    const unsigned long long mask = __ballot(hit);
    if (mask == 0ull) return;
    const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    if (lane_id() == leader) {
        const uint32_t wave = group * (blockDim.x >> 6) + (threadIdx.x >> 6);
        atomicAdd(&stats[wave % kStatShards].hits[level], (unsigned long long)__popcll(mask));
    }
}

// The work of one primary ray once it exists: trace, store the pixel's own lighting term,
// queue the children.
template <typename R, uint32_t SF, bool CAM>
RT_DEV void primary_ray(const SceneDev<R> &S, const PrimScene<R> &P, V3<R> ro, V3<R> rd, uint32_t k, bool valid,
                        uint32_t remaining, R *__restrict__ fb, const QueueDev<R> &out_q, StatShard *__restrict__ stats)
{
    Node<R> nd;
    nd.hit = false;
    nd.has_reflect = nd.has_refract = false;
    nd.lighting = { R(0), R(0), R(0) };
    if (valid) {
        nd = trace_node<R, SF, CAM>(S, CAM ? &P : nullptr, ro, rd, remaining);
        R *px = fb + (size_t)k * 3;
        px[0] = nd.lighting.x;
        px[1] = nd.lighting.y;
        px[2] = nd.lighting.z;
    }
    count_hits(stats, blockIdx.x, 0, valid && nd.hit);
    const uint32_t sub = blockIdx.x & out_q.sub_mask;
    if constexpr ((SF & SF_REFLECTIVE) != 0)
        queue_push(out_q, sub, valid && nd.has_reflect, nd.over_point, nd.reflectv, nd.reflect_factor, k, stats);
    if constexpr ((SF & SF_TRANSPARENT) != 0)
        queue_push(out_q, sub, valid && nd.has_refract, nd.under_point, nd.refract_dir, nd.refract_factor, k, stats);
}

// Camera rays of rows [row0, row0 + n_rows), one pixel per lane.  Each wave takes one 64-pixel
// segment of a row, so its framebuffer stores cover one contiguous run; with bounding-sphere
// tests it takes an 8x8 tile instead (camera_blocks() in rt_launch.h).  There is no loop: the
// camera block is dead once the ray exists, which leaves the scalar register file to the
// object tables.  The intersection walk reads the per-frame primary tables.
#ifndef RTX_CAMERA_ATTR
#define RTX_CAMERA_ATTR
#endif
template <typename R, uint32_t SF>
__global__ void __launch_bounds__(kCamBlock) RTX_CAMERA_ATTR
k_primary_camera(CameraDev<R> cam, SceneDev<R> S, PrimScene<R> P, uint32_t row0, uint32_t n_rows, uint32_t pieces_per_row,
                 uint32_t remaining, R *__restrict__ fb, QueueDev<R> out_q, StatShard *__restrict__ stats)
{
    const uint32_t piece = __builtin_amdgcn_readfirstlane(blockIdx.x * (kCamBlock / 64) + (threadIdx.x >> 6));
    const uint32_t piece_row = piece / pieces_per_row;
    const uint32_t piece_col = piece - piece_row * pieces_per_row;
    uint32_t row, x;
    if constexpr ((SF & SF_CULL) != 0) {
        row = piece_row * 8u + ((threadIdx.x >> 3) & 7u);
        x = piece_col * 8u + (threadIdx.x & 7u);
    } else {
        row = piece_row;
        x = piece_col * 64u + (threadIdx.x & 63u);
    }
    const bool valid = row < n_rows && x < cam.hsize;
    V3<R> ro = { R(0), R(0), R(0) }, rd = ro;
    if (valid) ray_for_pixel(cam, x, row0 + row, ro, rd);
    primary_ray<R, SF, true>(S, P, ro, rd, row * cam.hsize + x, valid, remaining, fb, out_q, stats);
}

// Rays from a caller-supplied array (World::color_at batches), one per thread.
template <typename R, uint32_t SF>
__global__ void __launch_bounds__(kRayBlock<R>)
k_primary_rays(SceneDev<R> S, const double *__restrict__ rays, uint32_t n, uint32_t remaining, R *__restrict__ fb,
               QueueDev<R> out_q, StatShard *__restrict__ stats)
{
    const PrimScene<R> none = {};
    const uint32_t k = blockIdx.x * kRayBlock<R> + threadIdx.x;
    const bool valid = k < n;
    V3<R> ro = { R(0), R(0), R(0) }, rd = ro;
    if (valid) {
        const double *r = rays + (size_t)k * 6;
        ro = { R(r[0]), R(r[1]), R(r[2]) };
        rd = { R(r[3]), R(r[4]), R(r[5]) };
    }
    primary_ray<R, SF, false>(S, none, ro, rd, k, valid, remaining, fb, out_q, stats);
}

// First kernel of a frame: zero the queue counters and the hit counters, and let one thread
// per object write the primary-ray table entry of its slot (n_tables = 0 for ray batches).
template <typename R>
__global__ void __launch_bounds__(kBlock)
k_frame_begin(CameraDev<R> cam, SceneDev<R> S, PrimScene<R> P, uint32_t n_tables, uint32_t *__restrict__ zero_a,
              uint32_t n_a, uint32_t *__restrict__ zero_b, uint32_t n_b)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_a) zero_a[i] = 0u;
    if (i < n_b) zero_b[i] = 0u;
    if (i < n_tables) make_primary_record<R>(S, P, { cam.origin[0], cam.origin[1], cam.origin[2] }, i);
}

template <typename R> RT_DEV void fb_add(R *p, R v);
template <> RT_DEV void fb_add<double>(double *p, double v) { unsafeAtomicAdd(p, v); }
template <> RT_DEV void fb_add<float>(float *p, float v) { unsafeAtomicAdd(p, v); }

// One recursion level: one queued ray per thread.  blockIdx.y selects the sub-queue and the
// grid's x extent covers its worst-case length, so workgroups past the actual length leave at
// once.  The children go to sub-queue (x + y) mod n_sub: every output sub-queue is fed by
// exactly gridDim.x workgroups, which is the bound the host sizes cap_sub with.
template <typename R, uint32_t SF>
__global__ void __launch_bounds__(kRayBlock<R>)
k_bounce(SceneDev<R> S, QueueDev<R> in_q, QueueDev<R> out_q, uint32_t remaining, uint32_t level,
         R *__restrict__ fb, StatShard *__restrict__ stats)
{
    const uint32_t sub_in = blockIdx.y;
    uint32_t n = in_q.count[(size_t)sub_in * kCountStride];
    if (n > in_q.cap_sub) n = in_q.cap_sub;
    // With bounding-sphere tests a wave walks the union of the spheres its rays can hit, one
    // dependent fetch after another, and deep levels hold few, incoherent rays: their waves are
    // long chains on a mostly idle chip.  While a level is short enough that the whole chip has
    // lanes to spare, give every ray `spread` lanes' worth of room (one ray per `spread` lanes):
    // more, shorter walks side by side.  The grid covers the worst-case length, so the room is
    // there; the decision is uniform over the launch's y slice.
    uint32_t spread = 1u;
    if constexpr ((SF & SF_CULL) != 0) {
        const uint32_t budget = kSpreadLanes / gridDim.y;  // lanes per sub-queue when the chip is lightly loaded
        while (spread < 64u && (uint64_t)n * (spread * 2u) <= budget && (uint64_t)n * (spread * 2u) <= (uint64_t)gridDim.x * kRayBlock<R>)
            spread *= 2u;
    }
    if ((uint64_t)blockIdx.x * kRayBlock<R> >= (uint64_t)n * spread) return;
    const uint32_t lane_slot = blockIdx.x * kRayBlock<R> + threadIdx.x;
    const uint32_t k = lane_slot / spread;
    const bool valid = k < n && (lane_slot & (spread - 1u)) == 0u;
    Node<R> nd;
    nd.hit = false;
    nd.has_reflect = nd.has_refract = false;
    R weight = R(0);
    uint32_t pixel = 0;
    if (valid) {
        using V2 = typename Vec2T<R>::type;
        const V2 *src = reinterpret_cast<const V2 *>(in_q.rec) + ((size_t)sub_in * in_q.cap_sub + k) * 4;
        const V2 a = src[0], b = src[1], c = src[2], e = src[3];
        V3<R> ro = { a.x, a.y, b.x };
        V3<R> rd = { b.y, c.x, c.y };
        weight = e.x;
        pixel = pixel_bits_load<R>(e.y);
        nd = trace_node<R, SF, false>(S, nullptr, ro, rd, remaining);
        if (nd.hit) {
            R *px = fb + (size_t)pixel * 3;
            fb_add<R>(px + 0, nd.lighting.x * weight);
            fb_add<R>(px + 1, nd.lighting.y * weight);
            fb_add<R>(px + 2, nd.lighting.z * weight);
        }
    }
    count_hits(stats, blockIdx.x + blockIdx.y, level, valid && nd.hit);
    const uint32_t sub = (blockIdx.x + blockIdx.y) & out_q.sub_mask;
    if constexpr ((SF & SF_REFLECTIVE) != 0)
        queue_push(out_q, sub, valid && nd.has_reflect, nd.over_point, nd.reflectv, weight * nd.reflect_factor, pixel, stats);
    if constexpr ((SF & SF_TRANSPARENT) != 0)
        queue_push(out_q, sub, valid && nd.has_refract, nd.under_point, nd.refract_dir, weight * nd.refract_factor, pixel, stats);
}

template <typename R>
__global__ void __launch_bounds__(kBlock)
k_raygen(CameraDev<R> cam, uint32_t pix_begin, uint32_t n_pix, double *__restrict__ out)
{
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n_pix; k += stride) {
        const uint32_t p = pix_begin + k;
        const uint32_t W = cam.hsize;
        uint32_t y = (uint32_t)((double)p * cam.inv_hsize);
        int32_t x = (int32_t)(p - y * W);
        if (x < 0) { y -= 1; x += (int32_t)W; }
        else if ((uint32_t)x >= W) { y += 1; x -= (int32_t)W; }
        V3<R> ro, rd;
        ray_for_pixel(cam, (uint32_t)x, y, ro, rd);
        double *o = out + (size_t)k * 6;
        o[0] = ro.x; o[1] = ro.y; o[2] = ro.z;
        o[3] = rd.x; o[4] = rd.y; o[5] = rd.z;
    }
}

// src/scenes/canvas.rs:84-86: `(c * 255.999) as u8` -- truncation, saturating, NaN -> 0
template <typename R>
__global__ void __launch_bounds__(kBlock) k_quantize(const R *__restrict__ rgb, size_t n, uint8_t *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        R v = rgb[i] * R(255.999);
        uint8_t q;
        if (!(v == v) || v <= R(0)) q = 0;
        else if (v >= R(255)) q = 255;
        else q = (uint8_t)(int)v;
        out[i] = q;
    }
}

// Canvas::canvas_to_png / canvas_to_buffer (src/scenes/canvas.rs:80-104): one pixel per thread,
// three channels `(c * 255.999) as u8`, and with four bytes per pixel an alpha byte of 255.
// Canvas::canvas_to_ppm (canvas.rs:52-71) scales by 255.0 instead, casts to isize and clamps to
// 0..255.  Both casts truncate toward zero, saturate and send NaN to 0, so one routine serves.
template <typename R> RT_DEV uint32_t to_u8(R c, R scale)
{
    const R v = c * scale;
    if (!(v == v) || v <= R(0)) return 0u;
    if (v >= R(255)) return 255u;
    return (uint32_t)(int)v;
}
template <typename R>
__global__ void __launch_bounds__(kBlock)
k_pack8(const R *__restrict__ rgb, uint32_t n_pix, uint32_t format, uint8_t *__restrict__ out)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n_pix) return;
    const R scale = (format >> 8) ? R(255.0) : R(255.999);
    const R *c = rgb + (size_t)p * 3;
    const uint32_t r = to_u8(c[0], scale), g = to_u8(c[1], scale), b = to_u8(c[2], scale);
    if ((format & 0xffu) == 4u) {
        reinterpret_cast<uint32_t *>(out)[p] = r | (g << 8) | (b << 16) | (255u << 24);
    } else {
        uint8_t *o = out + (size_t)p * 3;
        o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b;
    }
}

// ---- launchers --------------------------------------------------------------------------
// One instantiation per combination of the feature bits that change code.
// Scenes with groups share one general variant (reflection and refraction on, closest-hit shadow
// rays), with and without the bounding-sphere tests.
constexpr uint32_t sf_variant(uint32_t k)
{
    return k >= 16u ? (SF_REFLECTIVE | SF_TRANSPARENT | SF_CHAIN | ((k & 1u) ? SF_CULL : 0u))
                    : ((k & 7u) | ((k & 8u) ? SF_CULL : 0u));
}
constexpr uint32_t sf_case(uint32_t sf)
{
    return (sf & SF_CHAIN) ? (16u | ((sf & SF_CULL) ? 1u : 0u)) : ((sf & 7u) | ((sf & SF_CULL) ? 8u : 0u));
}
#define RTX_SF_SWITCH(sf, CALL)                                                \
    switch (sf_case(sf)) {                                                     \
    case 0u: CALL(sf_variant(0u)); break;                                      \
    case 1u: CALL(sf_variant(1u)); break;                                      \
    case 2u: CALL(sf_variant(2u)); break;                                      \
    case 3u: CALL(sf_variant(3u)); break;                                      \
    case 4u: CALL(sf_variant(4u)); break;                                      \
    case 5u: CALL(sf_variant(5u)); break;                                      \
    case 6u: CALL(sf_variant(6u)); break;                                      \
    case 7u: CALL(sf_variant(7u)); break;                                      \
    case 8u: CALL(sf_variant(8u)); break;                                      \
    case 9u: CALL(sf_variant(9u)); break;                                      \
    case 10u: CALL(sf_variant(10u)); break;                                    \
    case 11u: CALL(sf_variant(11u)); break;                                    \
    case 12u: CALL(sf_variant(12u)); break;                                    \
    case 13u: CALL(sf_variant(13u)); break;                                    \
    case 14u: CALL(sf_variant(14u)); break;                                    \
    case 15u: CALL(sf_variant(15u)); break;                                    \
    case 16u: CALL(sf_variant(16u)); break;                                    \
    default: CALL(sf_variant(17u)); break;                                     \
    }

template <typename R>
hipError_t Launch<R>::primary_camera(uint32_t sf, hipStream_t stream, const CameraDev<R> &cam, const SceneDev<R> &scene,
                                     const PrimScene<R> &prim, uint32_t row0, uint32_t n_rows, uint32_t remaining, R *fb,
                                     const QueueDev<R> &out_q, StatShard *stats)
{
    const uint32_t pieces_per_row = camera_pieces_per_row(sf, cam.hsize);
    const uint32_t grid = (uint32_t)camera_blocks(sf, cam.hsize, n_rows);
#define RTX_CALL(SFV)                                                                                               \
    hipLaunchKernelGGL((k_primary_camera<R, SFV | SF_PATTERN>), dim3(grid), dim3(kCamBlock), 0, stream, cam, scene, prim, \
                       row0, n_rows, pieces_per_row, remaining, fb, out_q, stats)
    RTX_SF_SWITCH(sf, RTX_CALL)
#undef RTX_CALL
    return hipGetLastError();
}

template <typename R>
hipError_t Launch<R>::primary_rays(uint32_t sf, hipStream_t stream, const SceneDev<R> &scene, const double *rays,
                                   uint32_t n, uint32_t remaining, R *fb, const QueueDev<R> &out_q, StatShard *stats)
{
    const uint32_t grid = (n + kRayBlock<R> - 1) / kRayBlock<R>;
#define RTX_CALL(SFV)                                                                                           \
    hipLaunchKernelGGL((k_primary_rays<R, SFV | SF_PATTERN>), dim3(grid), dim3(kRayBlock<R>), 0, stream, scene, rays, n, \
                       remaining, fb, out_q, stats)
    RTX_SF_SWITCH(sf, RTX_CALL)
#undef RTX_CALL
    return hipGetLastError();
}

template <typename R>
hipError_t Launch<R>::frame_begin(hipStream_t stream, const CameraDev<R> *cam, const SceneDev<R> &scene,
                                  const PrimScene<R> &prim, uint32_t *zero_a, uint32_t n_a, uint32_t *zero_b, uint32_t n_b)
{
    const uint32_t n_tables = cam ? scene.n : 0u;
    const uint32_t n = n_tables > n_a ? (n_tables > n_b ? n_tables : n_b) : (n_a > n_b ? n_a : n_b);
    if (n == 0) return hipSuccess;
    CameraDev<R> c = {};
    if (cam) c = *cam;
    hipLaunchKernelGGL((k_frame_begin<R>), dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, c, scene, prim,
                       n_tables, zero_a, n_a, zero_b, n_b);
    return hipGetLastError();
}

template <typename R>
hipError_t Launch<R>::bounce(uint32_t sf, hipStream_t stream, const SceneDev<R> &scene, const QueueDev<R> &in_q,
                             const QueueDev<R> &out_q, uint32_t remaining, uint32_t level, R *fb, StatShard *stats)
{
    if ((sf & (SF_REFLECTIVE | SF_TRANSPARENT)) == 0 || in_q.cap_sub == 0) return hipSuccess;  // nothing can be queued
    const dim3 grid((in_q.cap_sub + kRayBlock<R> - 1) / kRayBlock<R>, in_q.sub_mask + 1);
#define RTX_CALL(SFV)                                                                                   \
    if constexpr (((SFV) & (SF_REFLECTIVE | SF_TRANSPARENT)) != 0)                                      \
        hipLaunchKernelGGL((k_bounce<R, SFV | SF_PATTERN>), grid, dim3(kRayBlock<R>), 0, stream, scene, in_q, \
                           out_q, remaining, level, fb, stats)
    RTX_SF_SWITCH(sf, RTX_CALL)
#undef RTX_CALL
    return hipGetLastError();
}

template <typename R>
hipError_t Launch<R>::raygen(uint32_t grid, hipStream_t stream, const CameraDev<R> &cam, uint32_t pix_begin,
                             uint32_t n_pix, double *out)
{
    hipLaunchKernelGGL((k_raygen<R>), dim3(grid), dim3(kBlock), 0, stream, cam, pix_begin, n_pix, out);
    return hipGetLastError();
}

template <typename R>
hipError_t Launch<R>::quantize(uint32_t grid, hipStream_t stream, const R *rgb, size_t n, uint8_t *out)
{
    hipLaunchKernelGGL((k_quantize<R>), dim3(grid), dim3(kBlock), 0, stream, rgb, n, out);
    return hipGetLastError();
}

template <typename R>
hipError_t Launch<R>::pack8(hipStream_t stream, const R *rgb, uint32_t n_pix, uint32_t format, uint8_t *out)
{
    if (n_pix == 0) return hipSuccess;
    hipLaunchKernelGGL((k_pack8<R>), dim3((n_pix + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, rgb, n_pix, format, out);
    return hipGetLastError();
}

}  // namespace rtx
