// rt_launch.h -- host-callable launchers of the render kernels, one set per scalar type.
// The definitions live in render_f64.hip / render_f32.hip; the C ABI calls only these.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/rtx.h"
#include "rt_types.h"

namespace rtx {

// Threads per workgroup of the utility kernels.
constexpr uint32_t kBlock = 256;
// Threads per workgroup of the kernels that take rays from a queue or an array, one ray per
// thread without a loop (which keeps loop-carried state out of the scalar registers the object
// records need).  Measured on the glass and 10k-object scenes, f64 rays (dearer per ray) are
// best with 128, f32 rays with 256.
template <typename R> constexpr uint32_t kRayBlock = sizeof(R) == 8 ? 128u : 256u;
// Rays per piece of a bounce launch (and threads per workgroup of the bounce kernel).  Scenes without
// bounding-sphere tests stage their shading table in LDS per workgroup and do better with 256 threads
// in f64 as well (4K glass frame 1.84 -> 1.78 ms); with the tests, whose waves differ widely in length,
// 256 is slower (10k-object frame 5.97 -> 6.4 ms).
template <typename R> constexpr uint32_t bounce_block(uint32_t sf)
{
    if (sf & SF_LANE_TREE) return kLaneTreeBlock;
    return sizeof(R) == 8 ? ((sf & SF_CULL) ? 128u : 256u) : 256u;
}
// Lane slots per block index of a bounce launch of a scene with the sphere hierarchy: the children of block b of
// sub-queue y go to sub-queue (b + y) mod n_sub, whatever workgroup or wave traces them, which is what bounds the
// next level's sub-queues.
// A block is a workgroup's piece, except in the kernel that hands its rays out a wave at a time: there a block
// is what one wave takes, so that the rays of a long sub-queue, and in a short level every single ray, spread
// over as many sub-queues of the next level (deep levels come from few pixels; with blocks of 1024 the few
// sub-queues those pixels feed kept all their descendants: 0.27 ms for a level of 3 500 rays that takes 0.06
// when they are dealt out evenly).
template <typename R> constexpr uint32_t bounce_feed(uint32_t sf)
{
    if (sf & SF_LANE_TREE) return 64u;
    return bounce_block<R>(sf);
}
// How the kernel that keeps the packed hierarchy in LDS deals out a sub-queue of n rays (capacity cap_sub, n_sub
// sub-queues): the lanes each ray gets -- 64 in a short level (one ray per wave, the lanes side by side on the
// hierarchy; n * 64 lane slots stay within the sub-queue's capacity in whole blocks, so the bound on the output
// sub-queues holds), otherwise one.  The host repeats this when it sizes a level from measured lengths.
// Below how many rays a level gets one ray per wave: measured on 4K frames of 10 002 objects, the 35 000-ray
// level of the opaque scene takes 0.38 ms lane by lane and 0.30 ms a wave per ray, its 3 500-ray level 0.27 and
// 0.06 ms; with 25 glass objects, whose deeper levels hold 60 000 - 90 000 rays each, the frame takes 3.5 ms with
// the threshold at 4 096, 4.3 at 32 768 and 7.7 at 131 072: there the container walk of a wave per ray is made
// by 64 lanes that hold the same ray.
constexpr uint32_t kTreeWideRays = 65536u, kTreeWideRaysGlass = 4096u;
constexpr uint32_t tree_lanes_per_ray(uint32_t sf, uint32_t n, uint32_t cap_sub, uint32_t n_sub)
{
    if (n > cap_sub) n = cap_sub;
    const uint32_t feed = 64u;  // bounce_feed of this kernel
    const uint64_t room = ((uint64_t)cap_sub + feed - 1) / feed * feed;
    const uint32_t wide_below = (sf & SF_TRANSPARENT) ? kTreeWideRaysGlass : kTreeWideRays;
    return ((uint64_t)n * n_sub <= wide_below && (uint64_t)n * 64u <= room) ? 64u : 1u;
}
// One instantiation of the kernels per combination of the feature bits that change code: sf_variant(sf_case(sf))
// is the combination a scene with feature bits sf runs as.  Scenes with groups share one general variant
// (reflection and refraction on, closest-hit shadow rays), with and without the bounding-sphere tests.
constexpr uint32_t sf_variant(uint32_t k)
{
    return k >= 16u ? (SF_REFLECTIVE | SF_TRANSPARENT | SF_CHAIN | ((k & 1u) ? SF_CULL : 0u))
                    : ((k & 7u) | ((k & 8u) ? SF_CULL : 0u));
}
constexpr uint32_t sf_case(uint32_t sf)
{
    return (sf & SF_CHAIN) ? (16u | ((sf & SF_CULL) ? 1u : 0u)) : ((sf & 7u) | ((sf & SF_CULL) ? 8u : 0u));
}
// reflection and refraction: how many of the two the rays of a kernel variant can spawn (for a scene: of the
// variant it runs as, which is what lays out its regional queues)
constexpr uint32_t child_kinds(uint32_t sf) { return ((sf & SF_REFLECTIVE) ? 1u : 0u) + ((sf & SF_TRANSPARENT) ? 1u : 0u); }
constexpr uint32_t layout_kinds(uint32_t scene_sf) { return child_kinds(sf_variant(sf_case(scene_sf))); }
// Scenes walked object by object (no sphere hierarchy) keep the rays of a screen region, and of one kind, together.
// Sub-queues come in pairs: 2r holds the reflection rays and 2r + 1 the refraction rays that descend from the
// pixels of screen region r (a camera workgroup appends to the pair of the region it lies in, RegionMap; the
// children of a ray stay in the pair of their parent).  The 64 rays of a wave then start in the same part of the
// image, the same way, and mostly meet the same objects next, which is what the walk and the shading, written as
// one branch per object and case, need.  Sub-queues then differ widely in length, so the blocks of all of them
// are put in one list (k_list_blocks) and the workgroups of the level's launch take blocks from it in turn.
constexpr bool regional_levels(uint32_t sf) { return (sf & SF_CULL) == 0 && (sf & (SF_REFLECTIVE | SF_TRANSPARENT)) != 0; }
// bytes of a buffer that holds a queue of n_sub sub-queues of cap_sub entries (a multiple of 256: the next buffer
// starts behind it); behind the rays of a regional queue lies its block list
template <typename R> inline uint64_t queue_buffer_bytes(uint32_t sf, uint64_t entries, uint32_t n_sub)
{
    uint64_t b = entries * 8 * sizeof(R);
    if (regional_levels(sf)) b += (entries / kListBlock + n_sub + 1u) * 8u;  // every sub-queue ends with at most one partial block
    return (b + 255u) & ~(uint64_t)255u;
}
// `entries`: of all sub-queues together (n_sub * cap_sub, or what a queue sized from measured lengths holds)
template <typename R> inline void queue_list_plane(uint32_t sf, QueueDev<R> &q, uint64_t entries, bool var = false)
{
    q.list = nullptr;
    q.n_list = nullptr;
    q.plane = (uint32_t)entries;
    q.var = var ? 1u : 0u;
    if (regional_levels(sf) && q.rec) {
        q.list = reinterpret_cast<typename QueueDev<R>::Block *>(q.rec + entries * 8);
        q.n_list = q.count + 2;  // a spare word of the first sub-queue's counter line
    }
}
// Screen regions of a camera launch: nx x ny of them (powers of two), each `cols` workgroups wide and `rows`
// pixel rows high.  Row of workgroups `by` of the grid renders local row (by mod ny) * rows + by / ny: the
// workgroups resident at one time then lie in all ny rows of regions instead of one or two, and their
// appends go to as many counters (appends to one counter serialise).  The grid has ny * rows rows; those past
// the frame have nothing to do.  Workgroup bx of a row lies in region column bx / cols, taken as
// (bx * cols_magic) >> 32 (exact for bx, cols < 65536; cols_magic 0: cols is 1).
struct RegionMap {
    uint32_t cols_magic, nx, ny_shift, rows;
};
inline uint32_t region_magic(uint32_t d) { return d <= 1u ? 0u : (uint32_t)((((uint64_t)1 << 32) + d - 1u) / d); }
// n_regions (a power of two) regions for a grid: nx x ny of them, each cols x rows workgroups, about square in
// pixels (a workgroup is 64 pixels wide and one high)
inline void region_split(uint32_t grid_x, uint64_t grid_y, uint32_t n_regions, uint32_t *nx, uint32_t *cols, uint32_t *rows)
{
    uint32_t x = 1;
    while (x < n_regions && (uint64_t)x * x * grid_y < (uint64_t)n_regions * grid_x * 64u && x * 2u <= grid_x) x *= 2u;
    const uint32_t y = n_regions / x;
    *nx = x;
    *cols = (grid_x + x - 1u) / x;
    *rows = (uint32_t)((grid_y + y - 1u) / y);
}
// Threads per workgroup of the camera kernel.  Its waves share nothing.  In f64 one wave per
// workgroup: single waves fill the chip's last round more evenly than groups of four (4K headline
// frame: 61 -> 58.5 us).  The chip starts about 4.6 workgroups per nanosecond whatever their size
// (tools/wave_rate.hip: 28 us for the 129 600 one-wave workgroups of a 4K frame that do nothing, 8 us
// for 32 400 of four waves), which a frame of short f32 waves runs into.
template <typename R> constexpr uint32_t kCamBlock = 64u;

// Wave-sized pieces of the camera kernel's image for n_rows rows of hsize pixels.  Without
// bounding-sphere tests a wave takes one 64-pixel row segment (its framebuffer stores are one
// contiguous run).  With them it takes an 8x8 tile: the walk visits every sphere that any of
// the wave's rays can hit, and a square bundle of rays touches fewer than a flat one.
inline uint32_t camera_pieces_per_row(uint32_t sf, uint32_t hsize)
{
    return (sf & SF_CULL) ? (hsize + 7u) / 8u : (hsize + 63u) / 64u;
}
// The camera launch is a two-dimensional grid: blockIdx.y = row of pieces (at most 65 535 of them),
// blockIdx.x = workgroup within the row.
inline uint64_t camera_piece_rows(uint32_t sf, uint64_t n_rows) { return (sf & SF_CULL) ? (n_rows + 7u) / 8u : n_rows; }
template <typename R> inline uint32_t camera_blocks_per_row(uint32_t sf, uint32_t hsize)
{
    return (camera_pieces_per_row(sf, hsize) + kCamBlock<R> / 64 - 1) / (kCamBlock<R> / 64);
}
constexpr uint64_t kCameraMaxRows = 65535u - 1024u;  // rows of one camera launch: its grid numbers at most 65 535, of which a launch with screen regions leaves up to one per row of regions unused (of 8x8 tiles even more would do)
template <typename R> inline uint64_t camera_blocks(uint32_t sf, uint32_t hsize, uint64_t n_rows)
{
    return (uint64_t)camera_blocks_per_row<R>(sf, hsize) * camera_piece_rows(sf, n_rows);
}

template <typename R>
struct Launch {
    // Camera rays of local rows [local0, local0 + n_rows) of `map`: intersect, shade, store the
    // lighting term of pixel (row, x) at pixel index row * hsize + x of fb (row relative to local0;
    // format 0: three values of R per pixel, otherwise an 8-bit RTX_PIXELS_* image), queue the children.
    static hipError_t primary_camera(uint32_t sf, hipStream_t stream, const CameraDev<R> &cam, const SceneDev<R> &scene,
                                     const PrimScene<R> &prim, const RowMap &map, uint32_t local0, uint32_t n_rows,
                                     uint32_t remaining, void *fb, uint32_t format, const QueueDev<R> &out_q, StatShard *stats);
    // The same for rays[0..n) supplied by the caller (6 doubles each), result at fb[3 * k].
    static hipError_t primary_rays(uint32_t sf, hipStream_t stream, const SceneDev<R> &scene, const double *rays,
                                   uint32_t n, uint32_t remaining, R *fb, const QueueDev<R> &out_q, StatShard *stats);
    // First kernel of every frame: clears the two counter arrays and, for camera renders
    // (cam != nullptr), fills the per-frame primary-ray tables `prim` (device memory sized by
    // the caller) from the scene tables and the camera origin.
    static hipError_t frame_begin(hipStream_t stream, const CameraDev<R> *cam, const SceneDev<R> &scene,
                                  const PrimScene<R> &prim, uint32_t *zero_a, uint32_t n_a, uint32_t *zero_b,
                                  uint32_t n_b);
    // Primary rays only (no tracing): out[6*k..] = origin, direction.
    static hipError_t raygen(uint32_t grid, hipStream_t stream, const CameraDev<R> &cam, uint32_t pix_begin,
                             uint32_t n_pix, double *out);
    // 8-bit pixels from 3 values each; format: an RTX_PIXELS_* value (bytes per pixel in the low byte,
    // conversion rule above it)
    static hipError_t pack8(hipStream_t stream, const R *rgb, uint32_t n_pix, uint32_t format, uint8_t *out);
    // (c * 255.999) as u8
    static hipError_t quantize(uint32_t grid, hipStream_t stream, const R *rgb, size_t n, uint8_t *out);
};

// One recursion level below the primary rays: consume in_q, add weight * lighting into fb, queue
// the next level.  The launch has about resident_blocks workgroups (several times what the chip holds at
// once), shared out evenly among the sub-queues, or sharing the queue's block list in a scene with regional
// levels; they loop over the rays that exist.  `longest`:
// the length of the longest input sub-queue when the caller knows it (0: only its capacity is known).
template <typename R>
hipError_t launch_bounce(uint32_t sf, hipStream_t stream, const SceneDev<R> &scene, const QueueDev<R> &in_q,
                         const QueueDev<R> &out_q, uint32_t remaining, uint32_t level, R *fb, StatShard *stats,
                         uint32_t resident_blocks, uint32_t longest = 0);
extern template hipError_t launch_bounce<double>(uint32_t, hipStream_t, const SceneDev<double> &, const QueueDev<double> &,
                                                 const QueueDev<double> &, uint32_t, uint32_t, double *, StatShard *, uint32_t, uint32_t);
extern template hipError_t launch_bounce<float>(uint32_t, hipStream_t, const SceneDev<float> &, const QueueDev<float> &,
                                                const QueueDev<float> &, uint32_t, uint32_t, float *, StatShard *, uint32_t, uint32_t);

// Size of the file-scope device array behind rtx_render_device_noalloc (scratch.hip), and its
// device address on the current device.
constexpr size_t kScratchBytes = (size_t)256 << 20;
hipError_t scratch_address(void **ptr, size_t *bytes);
// sub-queue lengths into host memory the device can write, then a sequence number (scratch.hip)
hipError_t launch_publish_counts(hipStream_t stream, const uint32_t *counts, uint32_t n_sub, uint32_t stride, uint32_t *host, uint32_t seq);

// the block list of a regional queue (scratch.hip); nothing to do for other queues.  next_lines: the counter lines
// of the queue of the level after, when that queue will be sized from this one's lengths (k_list_blocks)
hipError_t list_blocks_launch(hipStream_t stream, const uint32_t *count, uint32_t n_sub, uint32_t cap_sub, uint32_t var, void *list,
                              uint32_t *n_list, uint32_t *next_lines, uint32_t kinds);
template <typename R> inline hipError_t launch_list_blocks(hipStream_t stream, uint32_t sf, const QueueDev<R> &q, uint32_t *next_lines = nullptr)
{
    if (!q.list || q.cap_sub == 0u) return hipSuccess;
    return list_blocks_launch(stream, q.count, q.sub_mask + 1u, q.cap_sub, q.var, q.list, q.n_list, next_lines, layout_kinds(sf));
}

// f64 only: intermediate results of one color_at node per ray.
hipError_t launch_probe(uint32_t sf, uint32_t grid, hipStream_t stream, const SceneDev<double> &scene,
                        const double *rays, uint32_t n, rtx_probe *out);

// f64 only: counts operand sets on which the shared-reciprocal quotients differ from plain divisions.
hipError_t launch_check_div3(hipStream_t stream, uint64_t seed, uint32_t n, uint32_t narrow, unsigned long long *mismatches);
hipError_t launch_check_pow(hipStream_t stream, uint64_t seed, uint32_t n, unsigned long long *out);

extern template struct Launch<double>;
extern template struct Launch<float>;

}  // namespace rtx
