// rt_types.h -- plain data shared by the host-side C ABI and the device code:
// the repacked object tables, the camera block, the secondary-ray queue and the counters.
#pragma once

#include <stdint.h>

namespace rtx {

// Shape tags: src/shapes/objects.rs:14-20.  Pattern tags: src/surfaces/patterns.rs:13-20.
enum : uint32_t { SH_SPHERE = 0, SH_PLANE = 1, SH_CUBE = 2, SH_CYLINDER = 3, SH_CONE = 4 };
enum : uint32_t { PT_SOLID = 0, PT_STRIPE = 1, PT_GRADIENT = 2, PT_RING = 3, PT_CHECKER = 4, PT_TEST = 5 };

enum : uint32_t { OF_CLOSED = 1u, OF_CASTS_SHADOW = 2u, OF_CONTAINER = 4u };
// Per slot.  SLOT_DIAGONAL: the upper-left 3x3 of the object's inverse transform has zeros off its
// diagonal (an object that was only translated and scaled, the common case): taking a ray into
// object space is then a multiply per component, plus an add for a point.
enum : uint32_t { SLOT_DIAGONAL = 1u };

// Scene-wide feature bits; the kernels are compiled once per combination that changes code.
enum : uint32_t {
    SF_REFLECTIVE = 1u,   // some object has reflective != 0
    SF_TRANSPARENT = 2u,  // some object has transparency != 0
    SF_ALL_CAST = 4u,     // every object casts a shadow: shadow rays may stop at any hit
    SF_PATTERN = 8u,      // some object carries a non-solid pattern
    SF_CULL = 16u,        // enough bounded objects that testing their bounding spheres first pays
    SF_CHAIN = 32u,       // some object sits inside a Group
    SF_LANE_TREE = 64u,   // the packed copy of the sphere hierarchy fits a workgroup's LDS (QuantNode, below)
};

// ---- intersection tables -------------------------------------------------------------
// Objects are regrouped by shape so that each shape gets its own loop with no per-object
// dispatch.  Within a shape the World's order is kept; a "slot" is an object's position in the
// concatenation spheres | planes | cubes | cylinders | cones.  Every loop indexes its table
// with a wave-uniform index, so the records travel through the scalar cache into SGPRs.

// sphere, cube: rows 0..2 of Object.inverse_transform (src/shapes/objects.rs:7)
template <typename R>
struct alignas(sizeof(R) * 4) XformRec {
    R m[12];
};
// plane: only row 1 of the inverse is ever read (src/shapes/planes.rs:11-17)
template <typename R>
struct alignas(sizeof(R) * 4) PlaneRec {
    R m[4];
};
// cylinder, cone: inverse rows plus the Shape payload (src/shapes/objects.rs:18-19)
template <typename R>
struct alignas(sizeof(R) * 4) CappedRec {
    R m[12];
    R ymin, ymax;
    uint32_t closed;
    uint32_t pad[(2 * sizeof(R) - 4) / 4];
};

// Object inside Groups (src/shapes/groups.rs:68-84): the ray is taken through the inverse of
// every enclosing group, outermost first, one transform per level as the reference does (the
// matrices are not pre-multiplied: that would round differently), then through the inverse of
// the child as it was added.  Such objects are rare and come last in slot order; the shape is
// a field, not a table.
constexpr uint32_t kMaxChain = 3;  // nesting depth of groups
template <typename R>
struct alignas(sizeof(R) * 4) ChainRec {
    R m[kMaxChain + 1][12];  // depth group inverses, then the child's own: rows 0..2 each
    R ymin, ymax;
    uint32_t type, closed, depth;
    uint32_t pad[sizeof(R) == 8 ? 1 : 3];
};

// Bounding sphere of one object in world space, by slot.  r = +inf for planes, cones and
// cylinders without finite y limits; finite radii are inflated on the host so that a line
// that misses the sphere cannot produce an intersection in the exact test either.
template <typename R>
struct alignas(sizeof(R) * 4) BoundRec {
    R c[3];
    R r;
};
// When the bounding-sphere tests are on, the slots of a shape table are ordered by a k-d
// partition of the objects' centres whose cuts fall on multiples of kCluster (and of kCluster^2
// higher up): every kCluster consecutive slots are neighbours in space and share one more
// bounding sphere (a cluster), every kCluster consecutive clusters share a third (a group).
// A ray tests n / kCluster^2 group spheres, then the clusters of the groups it meets, then the
// members of the clusters it meets.
constexpr uint32_t kCluster = 16;
// The same hierarchy once more, 8 bytes a sphere: centre and radius in units of `step` from `base`
// (SceneDev::q_*), the centre rounded to the nearest unit and the radius rounded up by more than the
// centre can have moved, so that the packed sphere holds the stored one.  r = 0xffff: unbounded.
// Small enough (objects, clusters and groups of a scene of ten thousand objects: 86 KB) to sit in the
// LDS of a workgroup, where every lane can walk it for its own ray at LDS latency (lane_tree_walk).
struct QuantNode {
    uint16_t c[3];
    uint16_t r;
};
constexpr uint32_t kLaneTreeMaxNodes = 11776;  // 92 KB of the 160 KB a workgroup can have
constexpr uint32_t kLaneTreeBlock = 1024;      // threads of a workgroup that holds the packed hierarchy (one per CU: 4 waves per SIMD)

// ---- primary-ray tables -----------------------------------------------------------------
// Every primary ray starts at the camera, so each object's object-space ray origin and the
// terms of its intersection test that depend on the origin alone are the same for all pixels.
// They are evaluated once per frame and object, with the expressions the generic path uses,
// into these tables (same slot order as the intersection tables).
template <typename R>
struct alignas(sizeof(R) * 2) PrimSphereRec {
    R m[9];    // upper-left 3x3 of Object.inverse_transform (what a direction is multiplied by)
    R o[3];    // object-space origin
    R c;       // o.o - 1
    R pad;
};
template <typename R>
struct alignas(sizeof(R) * 2) PrimPlaneRec {
    R m[3];    // row 1, columns 0..2 of the inverse
    R neg_oy;  // -(object-space origin).y
};
template <typename R>
struct alignas(sizeof(R) * 2) PrimCubeRec {
    R m[9];
    R lo[3];   // -1 - origin (the tmin numerators of cubes.rs:32)
    R hi[3];   //  1 - origin (the tmax numerators of cubes.rs:33)
    R pad;
};
template <typename R>
struct alignas(sizeof(R) * 2) PrimCappedRec {
    R m[9];
    R o[3];
    R o2[3];   // 2 * origin
    R c;       // cylinder: ox^2 + oz^2 - 1; cone: ox^2 - oy^2 + oz^2
    R ymin, ymax;
    uint32_t closed;
    uint32_t pad[(2 * sizeof(R) - 4) / 4];
};
// Bounding sphere of an object, cluster or group as the camera sees it: the tests of a primary ray
// against it need the centre relative to the camera and a few numbers made from the distance and
// the radius, all the same for every pixel (make_primary_bound fills them; may_hit_primary reads the
// first six for one ray, beam_may_hit all of them for the rays of a whole wave).
template <typename R>
struct alignas(sizeof(R) * 2) PrimBoundRec {
    R m[3];    // centre - camera origin
    R k;       // (1 - allowance) * |m|^2 - r^2: the line misses the sphere when k - (m.u)^2 > 0
    R r;       // radius (+inf: unbounded)
    R behind;  // the sphere lies behind the camera when m.u < behind: 0 with the camera outside it, else -inf
    R km;      // (1 - 4 allowance) * |m|^2: km - (m.a)^2 is a lower bound of the squared distance of the centre from axis a
    R dist;    // a lower bound of |m|
};
// the spheres of the hierarchy relative to one point (same indices as SceneDev::bnd, cl, grp)
template <typename R>
struct ApexBounds {
    const PrimBoundRec<R> *bnd, *cl, *grp;
};
template <typename R>
struct PrimScene {
    const PrimSphereRec<R> *sph;
    const PrimPlaneRec<R> *pln;
    const PrimCubeRec<R> *cub;
    const PrimCappedRec<R> *cyl;
    const PrimCappedRec<R> *con;
    uint32_t n_sph, n_pln, n_cub, n_cyl, n_con;
    uint32_t pad;
    // Scenes with bounding-sphere tests: for every shape table, its groups in the order of their distance
    // from the camera (entry k of a table's run is the table-relative index of the k-th nearest group;
    // the run of shape s starts at SceneDev::grp_off[s]).  Primary rays meet the groups in that order,
    // so that what they hit nearby rules out the groups farther away before those are entered.
    const uint32_t *grp_order;
    ApexBounds<R> eye;    // relative to the camera: for the camera rays
    ApexBounds<R> light;  // relative to the light: for the shadow rays of their hits
    // the shadow rays of a wave are walked together when the sine of their cone's half-angle is below this
    R shadow_beam_sin;
};

// Shading record: read once per shaded hit, gathered per lane by original object index.
template <typename R>
struct alignas(16) ColdRec {
    R inv[12];   // rows 0..2 of Object.inverse_transform
    R inv_t[9];  // rows 0..2, cols 0..2 of Object.inverse_transform_transposed (objects.rs:8)
    R ymin, ymax;
    R color[3];
    R ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
    R pinv[12];  // rows 0..2 of Pattern.inverse_transform (patterns.rs:7)
    R c1[3], c2[3];
    uint32_t type;
    uint32_t flags;
    uint32_t pattern_type;
    uint32_t pad;
};

template <typename R>
struct SceneDev {
    const XformRec<R> *sph;
    const PlaneRec<R> *pln;
    const XformRec<R> *cub;
    const CappedRec<R> *cyl;
    const CappedRec<R> *con;
    const ChainRec<R> *chn;    // objects inside groups: slots n - n_chn .. n - 1
    const BoundRec<R> *bnd;    // by slot
    const BoundRec<R> *cl;     // cluster spheres; those of shape s start at cl_off[s]
    const BoundRec<R> *grp;    // group spheres; those of shape s start at grp_off[s]
    uint32_t cl_off[5];
    uint32_t grp_off[5];
    const uint32_t *slot_obj;  // slot -> index in World.objects (the tie-break order among equal t)
    const uint32_t *slot_flags;  // by slot: SLOT_* bits
    const ColdRec<R> *cold;    // by index in World.objects
    uint32_t n_sph, n_pln, n_cub, n_cyl, n_con;
    uint32_t n;
    uint32_t flags;
    uint32_t n_chn;
    uint32_t n_grp_total;      // entries of grp (all shape tables)
    uint32_t n_bnd_total, n_cl_total;  // entries of bnd and of cl, padding included
    R light_pos[3];
    R light_int[3];
    R world_c[3];   // sphere around every object that has a finite bounding sphere (bounding-sphere
    R world_r;      // tests only; +inf: none)
    // SF_LANE_TREE: the packed hierarchy, nodes in the order bnd | cl | grp (same indices as those tables)
    const QuantNode *q_nodes;
    uint32_t q_n;
    uint32_t q_lds;  // where the running workgroup keeps its copy in LDS (set by the kernel that stages it)
    float q_base[3], q_step;
};

template <typename R>
struct CameraDev {
    R inv[12];     // rows 0..2 of Camera.inverse_transform (camera.rs:11)
    R origin[3];   // inverse_transform * point(0,0,0), evaluated once on the host (camera.rs:154)
    R pixel_size, half_width, half_height;
    uint32_t hsize, vsize;
    double inv_hsize;
};

// Which image rows a render call covers, in the order its output buffer holds them ("local"
// rows).  A call covers the bands of `band` rows that fall to part `part` of `n_parts` when bands
// are dealt out in turn (band b goes to part b mod n_parts), starting at image row `first`:
// local row L is image row first + ((L / band) * n_parts + part) * band + L % band.  A plain row
// range is the case n_parts = 1.
struct RowMap {
    uint32_t first, band, n_parts, part;
};

// Secondary-ray queue: 8 scalars per entry (origin, direction, path weight, pixel index).
// A queue is split into n_sub sub-queues of cap_sub entries, each with its own length counter:
// appends to ONE counter from every wave of the chip serialise (about 11 ns each on MI355X),
// appends spread over >= 256 counters do not.  A workgroup appends to the sub-queue its index
// selects, so the worst-case fill of every sub-queue is known on the host.
constexpr uint32_t kCountStride = 32;  // words between two counters: one 128-byte line each
template <typename R>
struct QueueDev {
    R *rec;            // sub-queue q occupies entries [q * cap_sub, (q + 1) * cap_sub)
    uint32_t *count;   // length of sub-queue q at count[q * kCountStride]; the word after it is the ticket counter of the
                       // launch that consumes the sub-queue (lane slots handed out so far; k_bounce_tree); both zero when a frame starts
    uint32_t cap_sub;  // entries per sub-queue (0: nothing may be appended)
    uint32_t sub_mask; // n_sub - 1, n_sub a power of two
    uint32_t *overflow; // set to 1 when an append finds its sub-queue full (host memory the device can write)
    // Queues of scenes whose levels keep the rays of a screen region together (regional_levels(), rt_launch.h;
    // null otherwise): the blocks of kListBlock entries of all sub-queues ({sub-queue, block within it},
    // *n_list of them, written by k_list_blocks between the producer and the consumer), from which the
    // workgroups of the consuming launch take their work in turn.
    struct Block { uint32_t sub, block; };
    Block *list;
    uint32_t *n_list;
    // Regional queues only.  plane: entries of all sub-queues together (what lies between two planes of the
    // layout).  var != 0: the queue was sized from the measured lengths of the level before; its sub-queues
    // differ in capacity (cap_sub is the largest) and lie back to back: sub-queue q begins at entry
    // count[q * kCountStride + kLineBegin] and holds at most count[q * kCountStride + kLineCap] entries
    // (written by k_list_blocks when it lists the level before).  var == 0: sub-queue q begins at q * cap_sub.
    uint32_t plane;
    uint32_t var;
};
constexpr uint32_t kLineBegin = 3, kLineCap = 4;  // words of a sub-queue's counter line (0: length, 1: ticket, 2: n_list in line 0)
constexpr uint32_t kListBlock = 256;   // entries per block of the list (= threads of the consuming workgroup)
constexpr uint32_t kListMaxSub = 512;  // most sub-queues of such a queue (256 screen regions of a scene that reflects and refracts;
                                       // 128 to 1024 regions measured alike on a 3840x2160 frame)

// Per-level counters, one block of shards per render call (summed on the host).
// lanes of a lightly loaded chip: 1024 SIMDs x 4 waves x 64 (k_bounce spreads short levels over them)
constexpr uint32_t kSpreadLanes = 1024u * 4u * 64u;
// camera rays of scenes with the sphere hierarchy are walked a wave at a time (beam_walk_table)
constexpr bool kBeamWalk = true;
constexpr int kStatShards = 256;
struct StatShard {
    unsigned long long hits[8];  // shaded hits (= is_shadowed calls) per recursion level
    unsigned long long pad[8];
};

}  // namespace rtx
