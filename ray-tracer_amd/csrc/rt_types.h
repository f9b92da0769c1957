// rt_types.h -- plain data shared by the host-side C ABI and the device code:
// the repacked object table, the camera block, the secondary-ray queue and the counters.
#pragma once

#include <stdint.h>

namespace rtx {

// ---- object table -------------------------------------------------------------------
// Shape tags: src/shapes/objects.rs:14-20.  Pattern tags: src/surfaces/patterns.rs:13-20.
enum : uint32_t { SH_SPHERE = 0, SH_PLANE = 1, SH_CUBE = 2, SH_CYLINDER = 3, SH_CONE = 4 };
enum : uint32_t { PT_SOLID = 0, PT_STRIPE = 1, PT_GRADIENT = 2, PT_RING = 3, PT_CHECKER = 4, PT_TEST = 5 };

enum : uint32_t { OF_CLOSED = 1u, OF_CASTS_SHADOW = 2u, OF_CONTAINER = 4u };

// Scene-wide feature bits; the kernels are compiled once per combination that changes code.
enum : uint32_t {
    SF_REFLECTIVE = 1u,   // some object has reflective != 0
    SF_TRANSPARENT = 2u,  // some object has transparency != 0
    SF_ALL_CAST = 4u,     // every object casts a shadow: shadow rays may stop at any hit
    SF_PATTERN = 8u,      // some object carries a non-solid pattern
};

// "Hot" record: everything a ray-object test reads.  The object loop indexes it with a
// wave-uniform index, so it is fetched through the scalar cache into SGPRs.
template <typename R>
struct alignas(sizeof(R) * 16) HotRec {
    R inv[12];  // rows 0..2 of Object.inverse_transform (objects.rs:7)
    R ymin, ymax;
    uint32_t type;
    uint32_t flags;
};
// "Cold" record: read once per shaded hit, gathered per lane.
template <typename R>
struct alignas(16) ColdRec {
    R inv_t[9];  // rows 0..2, cols 0..2 of Object.inverse_transform_transposed (objects.rs:8)
    R color[3];
    R ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
    R pinv[12];  // rows 0..2 of Pattern.inverse_transform (patterns.rs:7)
    R c1[3], c2[3];
    uint32_t pattern_type;
    uint32_t pad;
};

template <typename R>
struct SceneDev {
    const HotRec<R> *hot;
    const ColdRec<R> *cold;
    uint32_t n;
    uint32_t flags;
    R light_pos[3];
    R light_int[3];
};

template <typename R>
struct CameraDev {
    R inv[12];     // rows 0..2 of Camera.inverse_transform (camera.rs:11)
    R origin[3];   // inverse_transform * point(0,0,0), evaluated once on the host (camera.rs:154)
    R pixel_size, half_width, half_height;
    uint32_t hsize, vsize;
    double inv_hsize;
};

// Secondary-ray queue: 8 scalars per entry (origin, direction, path weight, pixel index).
template <typename R>
struct QueueDev {
    R *rec;
    uint32_t *count;
    uint32_t capacity;
    uint32_t pad;
};

// Per-level counters, one block of shards per render call (summed on the host).
constexpr int kStatShards = 64;
struct StatShard {
    unsigned long long hits[8];  // shaded hits (= is_shadowed calls) per recursion level
    unsigned long long overflow;
    unsigned long long pad[7];
};

}  // namespace rtx
