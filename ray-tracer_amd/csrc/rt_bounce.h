// rt_bounce.h -- the kernel of one recursion level below the primary rays and its launcher.
// It lives in translation units of its own (bounce_f64.hip, bounce_f32.hip): the kernel is one
// loop around a whole color_at node, and those units are built without machine-level invariant
// code motion (build.sh), which would otherwise move the literal constants of the loop body in
// front of the loop at the cost of 46 vector registers; the other kernels keep that pass.
#pragma once

#include <algorithm>

#include "rt_kernels.h"

namespace rtx {

template <typename R> RT_DEV void fb_add(R *p, R v);
template <> RT_DEV void fb_add<double>(double *p, double v) { unsafeAtomicAdd(p, v); }
template <> RT_DEV void fb_add<float>(float *p, float v) { unsafeAtomicAdd(p, v); }

// One recursion level is one launch of one of three kernels: k_bounce_regional (scenes walked object by
// object), k_bounce (the sphere hierarchy in memory), k_bounce_tree (its packed copy in LDS).
// The object tables are read through constant-address-space pointers (load_uniform), which keeps
// them in scalar registers although the loop body stores to memory.
// Everything a bounce launch is given, as one kernel argument: the loop below reads it afresh
// from the kernel-argument segment in every iteration instead of carrying some fifty scalar
// registers of pointers and counts across the loop (carried, they push the object records out
// of the scalar register file: 165 vector registers and lane spills instead of 110 and none).
template <typename R>
struct BounceArgs {
    SceneDev<R> S;
    QueueDev<R> in_q, out_q;
    uint32_t remaining, level;
    R *fb;
    StatShard *stats;
};
// LDS of a bounce workgroup: per wave a slice of `stride` chunks of 64 x 16 bytes.  Wave `wave`
// asks for entry idx of sub-queue `sub` for every lane with `want` set; the entry goes from memory
// straight into the first chunks of the wave's slice (global_load_lds: no vector registers are
// held while it is in flight), 16 bytes per lane and instruction.
template <typename R, bool REGIONAL = false>
RT_DEV void entry_prefetch(QueueChunk *lds, uint32_t stride, uint32_t wave, bool want, const QueueDev<R> &q, uint32_t sub, uint32_t idx)
{
    if (want) {
        const QueueChunk *src = reinterpret_cast<const QueueChunk *>(q.rec);
        for (uint32_t j = 0; j < kEntryChunks<R>; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + chunk_index<R, REGIONAL>(q, j, sub, idx)),
                                             (__attribute__((address_space(3))) void *)(lds + (wave * stride + j) * 64u), 16, 0, 0);
    }
}
// ... and takes it out of LDS once it has arrived (each lane reads what was loaded for it).
template <typename R>
RT_DEV void entry_take(const QueueChunk *lds, uint32_t stride, uint32_t wave, uint32_t lane, bool want, V3<R> &o, V3<R> &d, R &weight,
                       uint32_t &pixel)
{
    if (want) {
        // The transfers memory -> LDS count as vector memory operations: all of them have landed when
        // that counter is zero.  Stated here and not left to the compiler, which orders an LDS read after
        // such a transfer only where it can see that the two may touch the same bytes (with other LDS traffic
        // to the same array in between it has read two of the four chunks early).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        QueueChunk c[kEntryChunks<R>];
        for (uint32_t j = 0; j < kEntryChunks<R>; ++j) c[j] = lds[(wave * stride + j) * 64u + lane];
        R v[8];
        __builtin_memcpy(v, c, sizeof v);
        o = { v[0], v[1], v[2] };
        d = { v[3], v[4], v[5] };
        weight = v[6];
        pixel = pixel_bits_load<R>(v[7]);
    }
}

// The loop's trip count is unknown to the compiler, which would otherwise move the literal
// constants of the body (34 registers for the coefficients of pow alone) in front of it and
// end up with 165 vector registers; the translation unit is therefore built without
// machine-level invariant code motion (build.sh).
//
// Scenes walked object by object (regional_levels()): one recursion level.  The workgroups take the blocks of
// the queue's list in turn (kListBlock consecutive entries of one sub-queue), and the children of a ray stay in
// the sub-queues of its screen region, reflection rays in one and refraction rays in the other.
// Order of one round of the loop, chosen so that no memory latency is waited for:
//   1. ask for the entries of the block after this one (memory -> LDS, nothing waits) and for the list item of
//      the block after that;
//   2. closest hit of this block's rays;
//   3. the child rays, appended to the next level's queue at once (one instruction reserves the places of both
//      kinds), which frees their registers for the shadow ray, whose walk covers the latency of
//      the stores; shadow ray and lighting;
//   4. take the next block's entries from LDS (they arrived long ago), then add this block's
//      weight * lighting to the framebuffer: the atomics are in flight while the next round
//      computes.
// COLD_LDS: the launch comes with cold_lds_bytes() of dynamic LDS and the scene has at most
// kColdLdsObjects objects (launch_bounce sees to both).
template <typename R, uint32_t SF, bool COLD_LDS>
__global__ void __launch_bounds__(kListBlock)
k_bounce_regional(BounceArgs<R> args_in_kernarg_segment)
{
    static_assert(regional_levels(SF) && bounce_block<R>(SF) == kListBlock, "a block of the list is a workgroup's piece");
    constexpr uint32_t kPiece = kListBlock;
    constexpr uint32_t kKinds = child_kinds(SF);
    constexpr uint32_t kStride = kEntryChunks<R>;
    __shared__ QueueChunk lds[(kPiece / 64u) * kStride * 64u];
    const RT_CONSTANT_AS BounceArgs<R> *kp =
        (const RT_CONSTANT_AS BounceArgs<R> *)__builtin_amdgcn_kernarg_segment_ptr();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if constexpr (COLD_LDS) stage_cold_table<R>(args_in_kernarg_segment.S, kPiece);  // ends with a barrier
    V3<R> ro = { R(0), R(0), R(0) }, rd = ro;
    R weight = R(0);
    uint32_t pixel = 0;
    // Which blocks this workgroup takes.  The list holds the blocks sub-queue by sub-queue, i.e. screen region by
    // screen region, and the workgroups resident at one time hold consecutive indices: taken in list order they
    // would all append to the few sub-queues of one part of the image at the same time (appends to one counter
    // serialise).  Turn t therefore takes block (t * p) mod n_list, p a prime that does not divide n_list: a
    // permutation of the list in which neighbours in time are far apart in the image.
    const uint32_t n_list = ro_load(args_in_kernarg_segment.in_q.n_list);
    if (blockIdx.x >= n_list) return;
    const uint32_t stride_prime = n_list % 1000003u != 0u ? 1000003u : 1000033u;
    uint32_t at_list = (uint32_t)(((unsigned long long)blockIdx.x * stride_prime) % n_list);
    const uint32_t at_step = (uint32_t)(((unsigned long long)gridDim.x * stride_prime) % n_list);
    auto after = [&](uint32_t at) { return at + at_step >= n_list ? at + at_step - n_list : at + at_step; };
    using Block = typename QueueDev<R>::Block;
    // this round's block and the next round's (the same block again where the workgroup has no more turns)
    Block item = load_uniform(args_in_kernarg_segment.in_q.list + at_list), item1 = item;
    at_list = after(at_list);
    if (blockIdx.x + gridDim.x < n_list) item1 = load_uniform(args_in_kernarg_segment.in_q.list + at_list);
    // whether this thread's entry of a block exists, and which it is
    auto holds = [](const QueueDev<R> &q, const Block &it, uint32_t &k) {
        uint32_t n = ro_load(q.count + (size_t)it.sub * kCountStride);
        const uint32_t cap = sub_capacity<R, true>(q, it.sub);
        if (n > cap) n = cap;
        k = it.block * kPiece + threadIdx.x;
        return k < n;
    };
    uint32_t k;
    bool valid = holds(args_in_kernarg_segment.in_q, item, k);
    // the first block of this workgroup: nothing was asked for in advance
    entry_prefetch<R, true>(lds, kStride, wave, valid, args_in_kernarg_segment.in_q, item.sub, k);
    entry_take<R>(lds, kStride, wave, lane, valid, ro, rd, weight, pixel);
    for (uint32_t ticket = blockIdx.x;; ticket += gridDim.x) {
        // the compiler must not know that kp is the same pointer as in the last round, or it
        // hoists every load from it out of the loop (some fifty scalar registers)
        asm volatile("" : "+s"(kp));
        BounceArgs<R> A;
        __builtin_memcpy(&A, kp, sizeof A);
        const uint32_t region = item.sub / kKinds;
        const bool more = ticket + gridDim.x < n_list;  // wave-uniform
        uint32_t k_next = 0;
        const bool valid_next = more && holds(A.in_q, item1, k_next);
        entry_prefetch<R, true>(lds, kStride, wave, valid_next, A.in_q, item1.sub, k_next);
        at_list = after(at_list);
        Block item2 = item1;
        if (ticket + 2u * gridDim.x < n_list) item2 = load_uniform(A.in_q.list + at_list);

        Node<R> nd;
        nd.hit = false;
        nd.has_reflect = nd.has_refract = false;
        nd.lighting = { R(0), R(0), R(0) };
        const R w = weight;
        const uint32_t px_index = pixel;
        if (valid) {
            nd = trace_node<R, SF, false, WALK_WAVE, COLD_LDS>(A.S, nullptr, ro, rd, A.remaining, [&](const Node<R> &c) {
                queue_push_region(A.out_q, region, kKinds, c.has_reflect, c.over_point, c.reflectv, w * c.reflect_factor, c.has_refract,
                                  c.under_point, c.refract_dir, w * c.refract_factor, px_index);
            });
        }
        count_hits(A.stats, ticket, A.level, valid && nd.hit);
        entry_take<R>(lds, kStride, wave, lane, valid_next, ro, rd, weight, pixel);
        if (valid && nd.hit) {
            R *px = A.fb + (size_t)px_index * 3;
            fb_add<R>(px + 0, nd.lighting.x * w);
            fb_add<R>(px + 1, nd.lighting.y * w);
            fb_add<R>(px + 2, nd.lighting.z * w);
        }
        if (!more) break;
        valid = valid_next;
        item = item1;
        item1 = item2;
    }
}

// Scenes with the sphere hierarchy in memory: one recursion level.  blockIdx.y selects the input sub-queue; the
// gridDim.x workgroups of that row read its actual length on the device and share its rays, a piece
// (bounce_block rays) at a time (workgroup x takes blocks x, x + gridDim.x, ...): the launch has as many waves
// as the chip holds, whatever the worst-case capacity of the queue is.  The children of block b of sub-queue y
// go to output sub-queue (b + y) mod n_sub (deep levels come from few pixels, and their rays must spread over
// the chip), so every output sub-queue is fed by at most one block per block index: ceil(cap_sub /
// bounce_block) * bounce_block * children-per-ray entries, which is the bound the host sizes the next level's
// cap_sub with.  The entries of the next block are asked for (memory -> LDS) before this block's are traced.
constexpr uint32_t kWideRays = 65536u;  // a level of at most this many rays gets one ray per wave
template <typename R, uint32_t SF>
__global__ void __launch_bounds__(bounce_block<R>(SF))
k_bounce(BounceArgs<R> args_in_kernarg_segment)
{
    static_assert((SF & SF_CULL) != 0, "scenes with the sphere hierarchy");
    constexpr uint32_t kPiece = bounce_block<R>(SF);  // rays per piece = threads per workgroup
    constexpr uint32_t kStride = kEntryChunks<R>;
    __shared__ QueueChunk lds[(kPiece / 64u) * kStride * 64u];
    const RT_CONSTANT_AS BounceArgs<R> *kp =
        (const RT_CONSTANT_AS BounceArgs<R> *)__builtin_amdgcn_kernarg_segment_ptr();
    const uint32_t sub_in = blockIdx.y;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t step = gridDim.x * kPiece;
    V3<R> ro = { R(0), R(0), R(0) }, rd = ro;
    R weight = R(0);
    uint32_t pixel = 0;
    bool first = true;
    for (uint32_t base = blockIdx.x * kPiece;; base += step) {
        asm volatile("" : "+s"(kp));  // see k_bounce_regional
        BounceArgs<R> A;
        __builtin_memcpy(&A, kp, sizeof A);
        uint32_t n = ro_load(A.in_q.count + (size_t)sub_in * kCountStride);
        if (n > A.in_q.cap_sub) n = A.in_q.cap_sub;
        // A wave walks the union of the spheres its rays can hit, one
        // dependent fetch after another, and deep levels hold few, incoherent rays: their waves
        // are long chains on a mostly idle chip.  While a level is short enough that the whole
        // chip has lanes to spare, give every ray `spread` lanes' worth of room (one ray per
        // `spread` lanes): more, shorter walks side by side.  n * spread stays within the
        // sub-queue's capacity rounded up to whole blocks, so the block indices, and with them
        // the bound on the output sub-queues, do not grow.
        uint32_t spread = 1u;
        const uint32_t budget = kSpreadLanes / gridDim.y;  // lanes per sub-queue when the chip is lightly loaded
        const uint64_t room = ((uint64_t)A.in_q.cap_sub + kPiece - 1) / kPiece * kPiece;
        while (spread < 64u && (uint64_t)n * (spread * 2u) <= budget && (uint64_t)n * (spread * 2u) <= room) spread *= 2u;
        // A level of at most kWideRays rays gets a wave per ray, whose 64 lanes walk the sphere hierarchy
        // side by side (wide_walk_table): 64 waves per SIMD at most, each some 30 steps long.
        if ((uint64_t)n * gridDim.y <= kWideRays && (uint64_t)n * 64u <= room) spread = 64u;
        const bool wide = spread == 64u;  // wave-uniform
        const uint32_t total = n * spread;  // lane slots of this sub-queue
        if (base >= total) break;
        auto slot_of = [&](uint32_t block_base, uint32_t &k) {
            const uint32_t lane_slot = block_base + threadIdx.x;
            k = lane_slot / spread;
            return k < n && (lane_slot & (spread - 1u)) == 0u;
        };
        uint32_t k, k_next = 0;
        const bool valid = slot_of(base, k);
        if (first) {  // the first block of this workgroup: nothing was asked for in advance
            entry_prefetch<R>(lds, kStride, wave, valid, A.in_q, sub_in, k);
            entry_take<R>(lds, kStride, wave, lane, valid, ro, rd, weight, pixel);
            first = false;
        }
        const bool more = base + step < total;  // wave-uniform
        const bool valid_next = more && slot_of(base + step, k_next);
        entry_prefetch<R>(lds, kStride, wave, valid_next, A.in_q, sub_in, k_next);

        Node<R> nd;
        nd.hit = false;
        nd.has_reflect = nd.has_refract = false;
        nd.lighting = { R(0), R(0), R(0) };
        const uint32_t block = base / kPiece;
        const uint32_t sub = (block + sub_in) & A.out_q.sub_mask;
        const R w = weight;
        const uint32_t px_index = pixel;
        if (wide) {
            // the wave's one ray sits in lane 0 (the lane slots of a ray start at a multiple of 64):
            // every lane gets it, every lane computes the same node, lane 0 alone acts on it
            auto all = [](R v) {
                if constexpr (sizeof(R) == 8) {
                    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
                    return __builtin_bit_cast(R, ((unsigned long long)hi << 32) | lo);
                } else {
                    return __builtin_bit_cast(R, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
                }
            };
            if (__ballot(valid) != 0ull) {
                const V3<R> o = { all(ro.x), all(ro.y), all(ro.z) }, d = { all(rd.x), all(rd.y), all(rd.z) };
                nd = trace_node<R, SF, false, WALK_WIDE, false>(A.S, nullptr, o, d, A.remaining, [&](const Node<R> &c) {
                    queue_push2(A.out_q, sub, c.has_reflect && lane == 0u, c.over_point, c.reflectv, w * c.reflect_factor,
                                c.has_refract && lane == 0u, c.under_point, c.refract_dir, w * c.refract_factor, px_index);
                });
            }
        } else if (valid) {
            nd = trace_node<R, SF, false, WALK_WAVE, false>(A.S, nullptr, ro, rd, A.remaining, [&](const Node<R> &c) {
                queue_push2(A.out_q, sub, c.has_reflect, c.over_point, c.reflectv, w * c.reflect_factor, c.has_refract,
                            c.under_point, c.refract_dir, w * c.refract_factor, px_index);
            });
        }
        count_hits(A.stats, block + sub_in, A.level, valid && nd.hit);
        entry_take<R>(lds, kStride, wave, lane, valid_next, ro, rd, weight, pixel);
        if (valid && nd.hit) {
            R *px = A.fb + (size_t)px_index * 3;
            fb_add<R>(px + 0, nd.lighting.x * w);
            fb_add<R>(px + 1, nd.lighting.y * w);
            fb_add<R>(px + 2, nd.lighting.z * w);
        }
        if (!more) break;
    }
}

// One recursion level of a scene whose packed sphere hierarchy fits LDS (SF_LANE_TREE): every lane
// walks the hierarchy for its own ray (lane_tree_walk_table).  A workgroup copies the hierarchy into
// LDS once and then loops over its share of the sub-queue's rays, kLaneTreeBlock at a time, its
// waves running free of each other; the entries are read straight from the queue (the LDS is taken).
// The same queues, the same rule for where the children of a block go, as k_bounce.
template <typename R>
RT_DEV void entry_load(const QueueDev<R> &q, uint32_t sub, uint32_t idx, V3<R> &o, V3<R> &d, R &weight, uint32_t &pixel)
{
    const QueueChunk *src = reinterpret_cast<const QueueChunk *>(q.rec);
    QueueChunk c[kEntryChunks<R>];
    for (uint32_t j = 0; j < kEntryChunks<R>; ++j) c[j] = src[chunk_index(q, j, sub, idx)];
    R v[8];
    __builtin_memcpy(v, c, sizeof v);
    o = { v[0], v[1], v[2] };
    d = { v[3], v[4], v[5] };
    weight = v[6];
    pixel = pixel_bits_load<R>(v[7]);
}
constexpr uint32_t kTreeStealTries = 4u;  // a wave ends after it has used up this many sub-queues other than its own
template <typename R, uint32_t SF>
__global__ void __launch_bounds__(kLaneTreeBlock)
k_bounce_tree(BounceArgs<R> args_in_kernarg_segment)
{
    static_assert((SF & SF_CULL) != 0, "scenes with the sphere hierarchy");
    constexpr uint32_t kPiece = kLaneTreeBlock;
    constexpr uint32_t kFeed = bounce_feed<R>(SF | SF_LANE_TREE);
    static_assert(kFeed == 64u, "a wave takes one block of lane slots per ticket");
    __shared__ __attribute__((aligned(16))) unsigned long long tree[kLaneTreeMaxNodes];
    const RT_CONSTANT_AS BounceArgs<R> *kp =
        (const RT_CONSTANT_AS BounceArgs<R> *)__builtin_amdgcn_kernarg_segment_ptr();
    const uint32_t sub_in = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63u;
    {   // a workgroup without rays leaves before it copies anything
        uint32_t n = ro_load(args_in_kernarg_segment.in_q.count + (size_t)sub_in * kCountStride);
        if (n > args_in_kernarg_segment.in_q.cap_sub) n = args_in_kernarg_segment.in_q.cap_sub;
        if ((uint64_t)blockIdx.x * kPiece >= (uint64_t)n * 64u) return;
    }
    stage_lane_tree<R>(args_in_kernarg_segment.S, (RT_LDS_AS unsigned long long *)tree, kPiece);
    // Rays are handed out 64 lane slots at a time through a ticket counter per sub-queue (the word after its
    // length, zero when the frame starts): a wave that finishes a short walk takes the next ticket instead of
    // waiting for the slowest wave of a round, and a workgroup whose own sub-queue is used up goes on with its
    // neighbours'.  Which block a ray belongs to, and so where its children go, depends on its lane slot alone.
    const uint32_t n_sub = gridDim.y;
    // Lane slots of a sub-queue of n rays, and the lanes each ray gets (tree_lanes_per_ray).  (Levels between a
    // wave per ray and a lane per ray used to get one ray per 2 ... 16 lanes, as in k_bounce; with rays handed out
    // a wave at a time full waves are faster: 3.3 against 3.6 ms on the 4K frame of 10 002 objects with glass.)
    auto slots_of = [n_sub](uint32_t n, uint32_t cap_sub, uint32_t &spread) {
        if (n > cap_sub) n = cap_sub;
        spread = tree_lanes_per_ray(SF, n, cap_sub, n_sub);
        return n * spread;
    };
    uint32_t sub_cur = sub_in;
    uint32_t ahead = 0;  // sub-queues sub_in + 1 ... sub_in + ahead are known to be used up
    uint32_t others = 0;  // sub-queues other than its own that this wave went on with and found used up
    for (;;) {
        asm volatile("" : "+s"(kp));  // see k_bounce
        BounceArgs<R> A;
        __builtin_memcpy(&A, kp, sizeof A);
        A.S.q_lds = (uint32_t)(uintptr_t)(RT_LDS_AS unsigned long long *)tree;
        uint32_t *line = A.in_q.count + (size_t)sub_cur * kCountStride;
        uint32_t n = ro_load(line);
        if (n > A.in_q.cap_sub) n = A.in_q.cap_sub;
        uint32_t spread;
        const uint32_t total = slots_of(n, A.in_q.cap_sub, spread);
        const bool wide = spread == 64u;
        uint32_t base = 0;
        if (lane == 0u && total != 0u) base = atomicAdd(line + 1, kFeed);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (total == 0u || base >= total) {
            // This sub-queue is used up.  The 64 lanes look at the 64 sub-queues after the ones already known to be
            // used up (lengths and tickets read past the vector cache, which may hold what an earlier look saw);
            // the wave goes on with the first that has lane slots left, or ends when none has.  Not in a level of one
            // ray per wave, and not for ever: at the end of a level every wave would chase the last few tickets.
            if (sub_cur != sub_in) {
                ahead = ((sub_cur - sub_in) & (n_sub - 1u));  // it was the first with slots left: now it has none
                if (++others == kTreeStealTries) break;
            }
            if (total != 0u && wide) break;
            uint32_t found = 0;
            while (ahead + 1u < n_sub) {
                const uint32_t dist = ahead + 1u + lane;
                const bool in_range = dist < n_sub;
                const uint32_t *ln = A.in_q.count + (size_t)((sub_in + dist) & (n_sub - 1u)) * kCountStride;
                uint32_t ns = 0, tk = 0;
                if (in_range) {
                    ns = __hip_atomic_load(ln, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    tk = __hip_atomic_load(ln + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                uint32_t sp;
                const uint32_t slots = slots_of(ns, A.in_q.cap_sub, sp);
                const unsigned long long m = __ballot(in_range && tk < slots && sp != 64u);
                if (m != 0ull) {
                    found = ahead + 1u + (uint32_t)__builtin_ctzll(m);
                    break;
                }
                ahead += 64u;
            }
            if (found == 0u) break;
            ahead = found - 1u;
            sub_cur = (sub_in + found) & (n_sub - 1u);
            continue;
        }
        const uint32_t lane_slot = base + lane;
        const uint32_t k = lane_slot / spread;
        const bool valid = k < n && (lane_slot & (spread - 1u)) == 0u;
        Node<R> nd;
        nd.hit = false;
        nd.has_reflect = nd.has_refract = false;
        nd.lighting = { R(0), R(0), R(0) };
        const uint32_t block = base / kFeed;  // (a ticket is one block)
        const uint32_t sub = (block + sub_cur) & A.out_q.sub_mask;
        R w = R(0);
        uint32_t px_index = 0;
        V3<R> ro = { R(0), R(0), R(0) }, rd = ro;
        if (valid) entry_load<R>(A.in_q, sub_cur, k, ro, rd, w, px_index);
        if (wide) {
            auto all = [](R v) {
                if constexpr (sizeof(R) == 8) {
                    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
                    return __builtin_bit_cast(R, ((unsigned long long)hi << 32) | lo);
                } else {
                    return __builtin_bit_cast(R, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
                }
            };
            if (__ballot(valid) != 0ull) {  // the wave's one ray sits in lane 0
                const V3<R> o = { all(ro.x), all(ro.y), all(ro.z) }, d = { all(rd.x), all(rd.y), all(rd.z) };
                nd = trace_node<R, SF, false, WALK_TREE_WIDE, false>(A.S, nullptr, o, d, A.remaining, [&](const Node<R> &c) {
                    queue_push2(A.out_q, sub, c.has_reflect && lane == 0u, c.over_point, c.reflectv, w * c.reflect_factor,
                                c.has_refract && lane == 0u, c.under_point, c.refract_dir, w * c.refract_factor, px_index);
                });
            }
        } else if (valid) {
            nd = trace_node<R, SF, false, WALK_LANE, false>(A.S, nullptr, ro, rd, A.remaining, [&](const Node<R> &c) {
                queue_push2(A.out_q, sub, c.has_reflect, c.over_point, c.reflectv, w * c.reflect_factor, c.has_refract,
                            c.under_point, c.refract_dir, w * c.refract_factor, px_index);
            });
        }
        count_hits(A.stats, block + sub_cur, A.level, valid && nd.hit);
        if (valid && nd.hit) {
            R *px = A.fb + (size_t)px_index * 3;
            fb_add<R>(px + 0, nd.lighting.x * w);
            fb_add<R>(px + 1, nd.lighting.y * w);
            fb_add<R>(px + 2, nd.lighting.z * w);
        }
    }
}

template <typename R>
hipError_t launch_bounce(uint32_t sf, hipStream_t stream, const SceneDev<R> &scene, const QueueDev<R> &in_q,
                             const QueueDev<R> &out_q, uint32_t remaining, uint32_t level, R *fb, StatShard *stats,
                             uint32_t resident_blocks, uint32_t longest)
{
    if ((sf & (SF_REFLECTIVE | SF_TRANSPARENT)) == 0 || in_q.cap_sub == 0) return hipSuccess;  // nothing can be queued
    const uint32_t n_sub = in_q.sub_mask + 1;
    const uint32_t piece = bounce_block<R>(sf);
    const uint32_t worst = ((longest ? longest : in_q.cap_sub) + piece - 1) / piece;
    const uint32_t share = resident_blocks / n_sub > 0 ? resident_blocks / n_sub : 1u;
    const dim3 grid(worst < share ? worst : share, n_sub);
    const BounceArgs<R> args = { scene, in_q, out_q, remaining, level, fb, stats };
    if (sf & SF_LANE_TREE) {
        if (scene.q_n == 0 || scene.q_n > kLaneTreeMaxNodes || !scene.q_nodes) return hipErrorInvalidValue;
#define RTX_CALL(SFV)                                                                                   \
    if constexpr (((SFV) & (SF_REFLECTIVE | SF_TRANSPARENT)) != 0 && ((SFV) & SF_CULL) != 0)            \
        hipLaunchKernelGGL((k_bounce_tree<R, SFV | SF_PATTERN>), grid, dim3(piece), 0, stream, args);
        RTX_SF_SWITCH(sf, RTX_CALL)
#undef RTX_CALL
        return hipGetLastError();
    }
    if (regional_levels(sf)) {
        // one-dimensional: the workgroups share the queue's block list; scenes small enough keep their
        // shading table in LDS
        if (!in_q.list || !in_q.n_list) return hipErrorInvalidValue;
        const uint64_t most = (uint64_t)n_sub * (((longest ? longest : in_q.cap_sub) + piece - 1) / piece);
        const dim3 flat((uint32_t)std::min<uint64_t>(std::max<uint64_t>(most, 1u), resident_blocks));
        const uint32_t cold_bytes = cold_lds_bytes<R>(scene.n);
#define RTX_CALL(SFV)                                                                                   \
    if constexpr (regional_levels(SFV)) {                                                               \
        if (cold_bytes)                                                                                 \
            hipLaunchKernelGGL((k_bounce_regional<R, SFV | SF_PATTERN, true>), flat, dim3(piece), cold_bytes, stream, args); \
        else                                                                                            \
            hipLaunchKernelGGL((k_bounce_regional<R, SFV | SF_PATTERN, false>), flat, dim3(piece), 0, stream, args); \
    }
        RTX_SF_SWITCH(sf, RTX_CALL)
#undef RTX_CALL
        return hipGetLastError();
    }
#define RTX_CALL(SFV)                                                                                   \
    if constexpr (((SFV) & (SF_REFLECTIVE | SF_TRANSPARENT)) != 0 && ((SFV) & SF_CULL) != 0)            \
        hipLaunchKernelGGL((k_bounce<R, SFV | SF_PATTERN>), grid, dim3(piece), 0, stream, args);
    RTX_SF_SWITCH(sf, RTX_CALL)
#undef RTX_CALL
    return hipGetLastError();
}

}  // namespace rtx
