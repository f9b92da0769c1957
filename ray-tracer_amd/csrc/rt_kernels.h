// rt_kernels.h -- the render kernels and their launchers, templated on the scalar type.
//
// Kernel map (the reference's recursion, src/scenes/world.rs:121-190, unrolled by level):
//   k_primary  one pixel per lane: ray_for_pixel -> closest hit over the object table ->
//              shadow ray -> Phong; stores the pixel's own lighting term and appends the
//              reflection / refraction rays it spawns to the level-1 queue.
//   k_bounce_regional / k_bounce / k_bounce_tree (rt_bounce.h, by the kind of scene)
//              one queued ray per lane, same work; adds weight * lighting to its pixel and
//              appends to the next level's queue.  Launched once per recursion level.
// Queue appends are aggregated per wave: one atomic instruction per 64-lane wave reserves contiguous
// runs of slots, each lane writes its slot with 16-byte stores.
#pragma once

#include "rt_device.h"
#include "rt_launch.h"

namespace rtx {

template <typename R> struct Vec2T;
template <> struct Vec2T<double> { using type = double2; };
template <> struct Vec2T<float> { using type = float2; };

RT_DEV uint32_t region_div_dev(uint32_t v, uint32_t magic) { return magic ? __umulhi(v, magic) : v; }
RT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <typename R> RT_DEV uint32_t pixel_bits_load(R v);
template <> RT_DEV uint32_t pixel_bits_load<double>(double v) { return (uint32_t)__double_as_longlong(v); }
template <> RT_DEV uint32_t pixel_bits_load<float>(float v) { return __float_as_uint(v); }
template <typename R> RT_DEV R pixel_bits_store(uint32_t p);
template <> RT_DEV double pixel_bits_store<double>(uint32_t p) { return __longlong_as_double((long long)p); }
template <> RT_DEV float pixel_bits_store<float>(uint32_t p) { return __uint_as_float(p); }

// 16 bytes of a queue entry: the unit the entries travel in between memory, LDS and registers.
// A queue is laid out in planes: chunk j of entry idx of sub-queue q sits at chunk index
// (j * n_sub + q) * cap_sub + idx, so that the consecutive entries a wave appends or consumes
// are consecutive in memory for every one of its store and load instructions (one instruction
// moves 64 x 16 contiguous bytes; entry by entry it would touch 64 separate cache lines, and the
// address unit of the CU, which takes a line per cycle, becomes what the kernel waits for).
struct alignas(16) QueueChunk {
    uint32_t w[4];
};
template <typename R> constexpr uint32_t kEntryChunks = 8 * sizeof(R) / 16;
// REGIONAL: a queue of a scene with regional levels, whose sub-queues may differ in capacity (QueueDev::var);
// `sub` is then wave-uniform and the two words come through the scalar cache
template <typename R, bool REGIONAL = false> RT_DEV size_t chunk_index(const QueueDev<R> &q, uint32_t j, uint32_t sub, uint32_t idx)
{
    if constexpr (REGIONAL) {
        const size_t begin = q.var ? (size_t)ro_load(q.count + (size_t)sub * kCountStride + kLineBegin) : (size_t)sub * q.cap_sub;
        return (size_t)j * q.plane + begin + idx;
    } else {
        return ((size_t)j * (q.sub_mask + 1u) + sub) * q.cap_sub + idx;
    }
}
template <typename R, bool REGIONAL = false> RT_DEV uint32_t sub_capacity(const QueueDev<R> &q, uint32_t sub)
{
    if constexpr (REGIONAL) {
        if (q.var) return ro_load(q.count + (size_t)sub * kCountStride + kLineCap);
    }
    return q.cap_sub;
}
template <typename R, bool REGIONAL = false>
RT_DEV void queue_store(const QueueDev<R> &q, uint32_t sub, uint32_t idx, V3<R> o, V3<R> d, R weight, uint32_t pixel)
{
    if (idx < sub_capacity<R, REGIONAL>(q, sub)) {
        const R v[8] = { o.x, o.y, o.z, d.x, d.y, d.z, weight, pixel_bits_store<R>(pixel) };
        QueueChunk c[kEntryChunks<R>];
        __builtin_memcpy(c, v, sizeof v);
        QueueChunk *dst = reinterpret_cast<QueueChunk *>(q.rec);
        for (uint32_t j = 0; j < kEntryChunks<R>; ++j) dst[chunk_index<R, REGIONAL>(q, j, sub, idx)] = c[j];
    } else {
        // cannot happen (queues are sized for the worst case); if it does the ray is dropped and
        // the host reports the frame as failed
        __hip_atomic_store(q.overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// Append the children of one color_at node to sub-queue `sub` of the next level: the reflection
// ray (o1, d1, w1) where has1, the refraction ray (o2, d2, w2) where has2.  The lanes that call
// this together reserve all their slots with ONE atomic (issued by the first of them), the
// reflection rays first, then the refraction rays.
template <typename R>
RT_DEV void queue_push2(const QueueDev<R> &q, uint32_t sub, bool has1, V3<R> o1, V3<R> d1, R w1, bool has2, V3<R> o2, V3<R> d2,
                        R w2, uint32_t pixel)
{
    const unsigned long long m1 = __ballot(has1), m2 = __ballot(has2);
    if ((m1 | m2) == 0ull) return;
    const uint32_t lane = lane_id();
    const uint32_t leader = (uint32_t)__ffsll((long long)(m1 | m2)) - 1u;
    const uint32_t n1 = (uint32_t)__popcll(m1);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(q.count + (size_t)sub * kCountStride, n1 + (uint32_t)__popcll(m2));
    base = __shfl(base, (int)leader);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (has1) queue_store(q, sub, base + (uint32_t)__popcll(m1 & below), o1, d1, w1, pixel);
    if (has2) queue_store(q, sub, base + n1 + (uint32_t)__popcll(m2 & below), o2, d2, w2, pixel);
}

// The same for a regional queue (regional_levels()): the reflection rays go to the first and the refraction rays
// to the last of the `kinds` sub-queues of screen region `region` (two sub-queues in scenes that both reflect and
// refract, else one).  The two reservations travel in one instruction: the first lane that is here makes the one
// for the reflection rays, the second the one for the refraction rays (a lane alone makes both).
template <typename R>
RT_DEV void queue_push_region(const QueueDev<R> &q, uint32_t region, uint32_t kinds, bool has1, V3<R> o1, V3<R> d1, R w1, bool has2,
                              V3<R> o2, V3<R> d2, R w2, uint32_t pixel)
{
    const unsigned long long m1 = __ballot(has1), m2 = __ballot(has2);
    if ((m1 | m2) == 0ull) return;
    const uint32_t lane = lane_id();
    const unsigned long long here = __ballot(true);
    const uint32_t la = (uint32_t)__builtin_ctzll(here);
    const unsigned long long rest = here & (here - 1ull);
    const uint32_t lb = rest != 0ull ? (uint32_t)__builtin_ctzll(rest) : la;
    const uint32_t sub1 = region * kinds, sub2 = sub1 + kinds - 1u;
    const uint32_t n1 = (uint32_t)__popcll(m1), n2 = (uint32_t)__popcll(m2);
    uint32_t *const c1 = q.count + (size_t)sub1 * kCountStride, *const c2 = q.count + (size_t)sub2 * kCountStride;
    uint32_t got = 0u, got2 = 0u;
    if (lb != la) {
        if (lane == la || lane == lb) got = atomicAdd(lane == la ? c1 : c2, lane == la ? n1 : n2);
        got2 = (uint32_t)__shfl((int)got, (int)lb);
    } else if (lane == la) {
        got = atomicAdd(c1, n1);
        got2 = atomicAdd(c2, n2);
    }
    const uint32_t base1 = (uint32_t)__shfl((int)got, (int)la);
    const uint32_t base2 = lb != la ? got2 : (uint32_t)__shfl((int)got2, (int)la);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (has1) queue_store<R, true>(q, sub1, base1 + (uint32_t)__popcll(m1 & below), o1, d1, w1, pixel);
    if (has2) queue_store<R, true>(q, sub2, base2 + (uint32_t)__popcll(m2 & below), o2, d2, w2, pixel);
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

// Canvas::canvas_to_png / canvas_to_buffer (src/scenes/canvas.rs:80-104): three channels
// `(c * 255.999) as u8`, and with four bytes per pixel an alpha byte of 255.
// Canvas::canvas_to_ppm (canvas.rs:52-71) scales by 255.0 instead, casts to isize and clamps to
// 0..255.  Both casts truncate toward zero, saturate and send NaN to 0, so one routine serves.
template <typename R> RT_DEV uint32_t to_u8(R c, R scale)
{
    const R v = c * scale;
    if (!(v == v) || v <= R(0)) return 0u;
    if (v >= R(255)) return 255u;
    return (uint32_t)(int)v;
}
// Pixel p of an 8-bit image in `format` (an RTX_PIXELS_* value: bytes per pixel in the low byte,
// conversion rule above it).
template <typename R> RT_DEV void store_pixel8(uint8_t *__restrict__ out, uint32_t p, uint32_t format, R cr, R cg, R cb)
{
    const R scale = (format >> 8) ? R(255.0) : R(255.999);
    const uint32_t r = to_u8(cr, scale), g = to_u8(cg, scale), b = to_u8(cb, scale);
    if ((format & 0xffu) == 4u) {
        reinterpret_cast<uint32_t *>(out)[p] = r | (g << 8) | (b << 16) | (255u << 24);
    } else {
        uint8_t *o = out + (size_t)p * 3;
        o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b;
    }
}

// The work of one primary ray once it exists: trace, store the pixel's own lighting term,
// queue the children.  format 0: fb holds 3 values of R per pixel; otherwise it is an 8-bit
// image in that format and the term is converted as it is stored (only frames whose pixels
// receive nothing from deeper levels are rendered that way).
template <typename R, uint32_t SF, bool CAM, bool OUT8 = false>
RT_DEV void primary_ray(const SceneDev<R> &S, const PrimScene<R> &P, V3<R> ro, V3<R> rd, uint32_t k, bool valid,
                        uint32_t remaining, void *__restrict__ fb, uint32_t format, const QueueDev<R> &out_q,
                        StatShard *__restrict__ stats, uint32_t block_id)
{
    Node<R> nd;
    nd.hit = false;
    nd.has_reflect = nd.has_refract = false;
    nd.lighting = { R(0), R(0), R(0) };
    // (block_id of a scene with regional levels: the screen region, or any number for a ray batch)
    const uint32_t sub = block_id & out_q.sub_mask;
    auto children = [&](const Node<R> &c) {
        if constexpr (regional_levels(SF))
            queue_push_region(out_q, block_id & ((out_q.sub_mask + 1u) / child_kinds(SF) - 1u), child_kinds(SF), c.has_reflect, c.over_point,
                              c.reflectv, c.reflect_factor, c.has_refract, c.under_point, c.refract_dir, c.refract_factor, k);
        else if constexpr ((SF & (SF_REFLECTIVE | SF_TRANSPARENT)) != 0)
            queue_push2(out_q, sub, c.has_reflect, c.over_point, c.reflectv, c.reflect_factor, c.has_refract, c.under_point,
                        c.refract_dir, c.refract_factor, k);
    };
    // camera rays of a scene with the sphere hierarchy: the wave finds its closest hits together, and
    // then whether they are shadowed (shadow rays all end at the light)
    constexpr bool kBeam = CAM && (SF & SF_CULL) != 0 && kBeamWalk;
    if constexpr (kBeam) {
        Hit<R> h = closest_hit_primary_beam<R, (SF & SF_CHAIN) != 0>(S, P, ro, rd, valid);
        if (valid) containers_if_needed<R, SF, false>(S, ro, rd, remaining, h);
        ShadowQuery<R> query = { { R(0), R(0), R(0) }, R(0) };
        nd.over_point = { R(0), R(0), R(0) };
        if (valid) nd = shade_prepare<R, SF, CAM, false>(S, ro, rd, remaining, h, query, children);
        const bool asks = valid && nd.hit;
        bool shadowed = false;
        if constexpr ((SF & SF_ALL_CAST) != 0) shadowed = is_shadowed_beam<R, SF>(S, P, nd.over_point, query.lightv, query.ldist, asks);
        else if (asks) shadowed = is_shadowed<R, SF, WALK_WAVE, false>(S, nd.over_point, query.lightv, query.ldist);
        if (asks) shade_finish<R, SF, CAM, false>(S, rd, nd, query, shadowed);
    }
    if (valid) {
        if constexpr (!kBeam) nd = trace_node<R, SF, CAM, WALK_WAVE>(S, CAM ? &P : nullptr, ro, rd, remaining, children);
        if constexpr (!OUT8) {
            R *px = static_cast<R *>(fb) + (size_t)k * 3;
            px[0] = nd.lighting.x;
            px[1] = nd.lighting.y;
            px[2] = nd.lighting.z;
        } else {
            store_pixel8<R>(static_cast<uint8_t *>(fb), k, format, nd.lighting.x, nd.lighting.y, nd.lighting.z);
        }
    }
    count_hits(stats, block_id, 0, valid && nd.hit);
}

// Camera rays of local rows [local0, local0 + n_rows) of `map`, one pixel per lane; the output
// pixel of local row L and column x is L * hsize + x, counted from fb.  Each wave takes one 64-pixel
// segment of a row, so its framebuffer stores cover one contiguous run; with bounding-sphere
// tests it takes an 8x8 tile instead (camera_blocks() in rt_launch.h).  There is no loop: the
// camera block is dead once the ray exists, which leaves the scalar register file to the
// object tables.  The intersection walk reads the per-frame primary tables.
// Waves per SIMD the camera kernel is compiled for (the register allocator's ceiling follows from it).
// Left alone the compiler takes 80-90 vector registers for the f64 variants without bounding-sphere
// tests (5 waves per SIMD) although 66 do without a spill; their waves are short chains of dependent
// f64 operations behind two rounds of scalar loads, and more of them side by side is what hides that.
// The variants with the tests spill at 6 waves (40-130 bytes of scratch; camera launch of a 10 002-object
// frame 1.05 -> 1.16 ms) and keep the compiler's choice.
constexpr uint32_t camera_waves(uint32_t sf) { return (sf & SF_CULL) ? 1u : 7u; }
template <typename R, uint32_t SF, bool OUT8 = false, bool BANDED = false>
__global__ void __launch_bounds__(kCamBlock<R>) __attribute__((amdgpu_waves_per_eu(camera_waves(SF))))
k_primary_camera(CameraDev<R> cam, SceneDev<R> S, PrimScene<R> P, uint32_t local0, uint32_t n_rows,
                 uint32_t remaining, void *__restrict__ fb, QueueDev<R> out_q, StatShard *__restrict__ stats, RowMap map,
                 uint32_t format, RegionMap regions)
{
    // a two-dimensional grid: row of pieces = blockIdx.y (a division by pieces_per_row otherwise, thirty
    // instructions in front of everything else in waves of three hundred)
    uint32_t piece_row = blockIdx.y;
    const uint32_t piece_col = __builtin_amdgcn_readfirstlane(blockIdx.x * (kCamBlock<R> / 64) + (threadIdx.x >> 6));
    // which sub-queue the children go to: by workgroup in turn, or by the screen region the workgroup lies in
    uint32_t block_id = blockIdx.y * gridDim.x + blockIdx.x;
    if constexpr (regional_levels(SF)) {
        const uint32_t region_row = blockIdx.y & ((1u << regions.ny_shift) - 1u);
        piece_row = region_row * regions.rows + (blockIdx.y >> regions.ny_shift);
        block_id = region_row * regions.nx + region_div_dev(blockIdx.x, regions.cols_magic);
    }
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
    if (valid) {
        // plain row ranges come with local0 = their first image row and never read the map
        uint32_t image_row = local0 + row;
        if constexpr (BANDED)
            image_row = map.first + ((image_row / map.band) * map.n_parts + map.part) * map.band + image_row % map.band;
        ray_for_pixel(cam, x, image_row, ro, rd);
    }
    primary_ray<R, SF, true, OUT8>(S, P, ro, rd, row * cam.hsize + x, valid, remaining, fb, format, out_q, stats, block_id);
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
    primary_ray<R, SF, false>(S, none, ro, rd, k, valid, remaining, fb, 0u, out_q, stats, blockIdx.x);
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
    if (n_tables != 0u && i < S.n_grp_total) rank_group<R>(S, P, { cam.origin[0], cam.origin[1], cam.origin[2] }, i);
    if (n_tables != 0u && P.eye.bnd) {
        make_primary_bound<R>(S, P.eye, { cam.origin[0], cam.origin[1], cam.origin[2] }, i);
        make_primary_bound<R>(S, P.light, { S.light_pos[0], S.light_pos[1], S.light_pos[2] }, i);
    }
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

// One pixel per thread of an R-valued image into an 8-bit one (store_pixel8).
template <typename R>
__global__ void __launch_bounds__(kBlock)
k_pack8(const R *__restrict__ rgb, uint32_t n_pix, uint32_t format, uint8_t *__restrict__ out)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n_pix) return;
    const R *c = rgb + (size_t)p * 3;
    store_pixel8<R>(out, p, format, c[0], c[1], c[2]);
}

// ---- launchers --------------------------------------------------------------------------
// One instantiation per combination of the feature bits that change code (sf_variant, rt_launch.h).
#ifdef RTX_ONLY_SF
// tools builds (ISA inspection, experiments): a single kernel variant, whatever the scene asks for
#define RTX_SF_SWITCH(sf, CALL) { CALL((RTX_ONLY_SF)); }
#else
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
#endif

template <typename R>
hipError_t Launch<R>::primary_camera(uint32_t sf, hipStream_t stream, const CameraDev<R> &cam, const SceneDev<R> &scene,
                                     const PrimScene<R> &prim, const RowMap &map, uint32_t local0, uint32_t n_rows,
                                     uint32_t remaining, void *fb, uint32_t format, const QueueDev<R> &out_q, StatShard *stats)
{
    dim3 grid(camera_blocks_per_row<R>(sf, cam.hsize), camera_piece_rows(sf, n_rows));
    RegionMap regions = {};
    if (regional_levels(sf)) {
        const uint32_t n_regions = (out_q.sub_mask + 1u) / layout_kinds(sf);
        uint32_t cols;
        region_split(grid.x, grid.y, n_regions, &regions.nx, &cols, &regions.rows);
        regions.cols_magic = region_magic(cols);
        const uint32_t ny = n_regions / regions.nx;
        while ((1u << regions.ny_shift) < ny) ++regions.ny_shift;
        grid.y = ny * regions.rows;  // (at most ny - 1 rows more than the frame has)
    }
    if (grid.y > 65535u) return hipErrorInvalidValue;  // plan_frame keeps chunks below that
    // the 8-bit variants exist for the scenes that can use them: those in which nothing reflects or
    // refracts; the variants that deal bands out to parts spare the plain row range their divisions
    const bool banded = map.n_parts != 1u;
    const uint32_t first = banded ? local0 : map.first + local0;  // what the kernel adds its row to
#define RTX_LAUNCH_CAM(SFV, OUT8, BANDED)                                                                            \
    hipLaunchKernelGGL((k_primary_camera<R, SFV | SF_PATTERN, OUT8, BANDED>), grid, dim3(kCamBlock<R>), 0, \
                       stream, cam, scene, prim, first, n_rows, remaining, fb, out_q, stats, map, format, regions)
#define RTX_CALL(SFV)                                                                                                \
    if constexpr (((SFV) & (SF_REFLECTIVE | SF_TRANSPARENT)) == 0) {                                               \
        if (format != 0u) { if (banded) RTX_LAUNCH_CAM(SFV, true, true); else RTX_LAUNCH_CAM(SFV, true, false); }    \
        else { if (banded) RTX_LAUNCH_CAM(SFV, false, true); else RTX_LAUNCH_CAM(SFV, false, false); }               \
    } else {                                                                                                         \
        if (format != 0u) return hipErrorInvalidValue;                                                               \
        if (banded) RTX_LAUNCH_CAM(SFV, false, true); else RTX_LAUNCH_CAM(SFV, false, false);                        \
    }
    RTX_SF_SWITCH(sf, RTX_CALL)
#undef RTX_CALL
#undef RTX_LAUNCH_CAM
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
    const uint32_t n_tables = cam ? scene.n : 0u;  // (a scene has fewer groups than objects)
    const uint32_t n_bounds = (cam && prim.eye.bnd) ? scene.n_bnd_total + scene.n_cl_total + scene.n_grp_total : 0u;
    uint32_t n = n_tables > n_a ? (n_tables > n_b ? n_tables : n_b) : (n_a > n_b ? n_a : n_b);
    n = n > n_bounds ? n : n_bounds;
    if (n == 0) return hipSuccess;
    CameraDev<R> c = {};
    if (cam) c = *cam;
    hipLaunchKernelGGL((k_frame_begin<R>), dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, c, scene, prim,
                       n_tables, zero_a, n_a, zero_b, n_b);
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
