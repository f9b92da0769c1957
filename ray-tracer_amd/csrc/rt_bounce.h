// rt_bounce.h -- the kernel of one recursion level below the primary rays and its launcher.
// It lives in translation units of its own (bounce_f64.hip, bounce_f32.hip): the kernel is one
// loop around a whole color_at node, and those units are built without machine-level invariant
// code motion (build.sh), which would otherwise move the literal constants of the loop body in
// front of the loop at the cost of 46 vector registers; the other kernels keep that pass.
#pragma once

#include "rt_kernels.h"

namespace rtx {

template <typename R> RT_DEV void fb_add(R *p, R v);
template <> RT_DEV void fb_add<double>(double *p, double v) { unsafeAtomicAdd(p, v); }
template <> RT_DEV void fb_add<float>(float *p, float v) { unsafeAtomicAdd(p, v); }

// One recursion level.  blockIdx.y selects the input sub-queue; the gridDim.x workgroups of that
// row read its actual length on the device and share its rays, a piece (bounce_block rays) at a time (workgroup x
// takes blocks x, x + gridDim.x, ...): the launch has as many waves as the chip holds, whatever
// the worst-case capacity of the queue is.  The children of block b of sub-queue y go to output
// sub-queue (b + y) mod n_sub, so every output sub-queue is fed by at most one block per block
// index: ceil(cap_sub / bounce_block) * bounce_block * children-per-ray entries, which is the bound
// the host sizes the next level's cap_sub with.
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
// Wave `wave` of a workgroup asks for entry idx of sub-queue `sub` for every lane with `want`
// set; the entry goes from memory straight into the wave's slice of `lds` (global_load_lds: no
// vector registers are held while it is in flight), 16 bytes per lane and instruction.
template <typename R>
RT_DEV void entry_prefetch(QueueChunk *lds, uint32_t wave, bool want, const QueueDev<R> &q, uint32_t sub, uint32_t idx)
{
    if (want) {
        const QueueChunk *src = reinterpret_cast<const QueueChunk *>(q.rec);
        for (uint32_t j = 0; j < kEntryChunks<R>; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + chunk_index(q, j, sub, idx)),
                                             (__attribute__((address_space(3))) void *)(lds + (wave * kEntryChunks<R> + j) * 64u), 16, 0, 0);
    }
}
// ... and takes it out of LDS once it has arrived (each lane reads what was loaded for it).
template <typename R>
RT_DEV void entry_take(const QueueChunk *lds, uint32_t wave, uint32_t lane, bool want, V3<R> &o, V3<R> &d, R &weight, uint32_t &pixel)
{
    if (want) {
        QueueChunk c[kEntryChunks<R>];
        for (uint32_t j = 0; j < kEntryChunks<R>; ++j) c[j] = lds[(wave * kEntryChunks<R> + j) * 64u + lane];
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
// Order of one round of the loop, chosen so that no memory latency is waited for:
//   1. ask for the entries of the block after this one (memory -> LDS, nothing waits);
//   2. closest hit and the child rays of this block's rays; the children are appended to the
//      next level's queue at once (one atomic per wave for both kinds), which frees their
//      registers for the shadow ray, whose walk covers the latency of the stores;
//   3. shadow ray and lighting;
//   4. take the next block's entries from LDS (they arrived long ago), then add this block's
//      weight * lighting to the framebuffer: the atomics are in flight while the next round
//      computes.
#ifdef RTX_PHASE_CLOCK
static __device__ unsigned int g_slow_flag;
static __device__ unsigned long long g_slow_threshold = 3000000ull;  // cycles of a round above which its rays are kept
static __device__ double g_slow_rays[64 * 8];
#endif
// COLD_LDS: the launch comes with cold_lds_bytes() of dynamic LDS and the scene has at most
// kColdLdsObjects objects (launch_bounce sees to both).
#ifndef RTX_WIDE_RAYS  // a level of at most this many rays gets one ray per wave (scenes with the sphere hierarchy)
#define RTX_WIDE_RAYS 65536u
#endif
#ifndef RTX_BOUNCE_ATTR  // occupancy experiments, e.g. -DRTX_BOUNCE_ATTR='__attribute__((amdgpu_waves_per_eu(5,5)))'
#define RTX_BOUNCE_ATTR
#endif
template <typename R, uint32_t SF, bool COLD_LDS>
__global__ void __launch_bounds__(bounce_block<R>(SF)) RTX_BOUNCE_ATTR
k_bounce(BounceArgs<R> args_in_kernarg_segment)
{
    constexpr uint32_t kPiece = bounce_block<R>(SF);  // rays per piece = threads per workgroup
    __shared__ QueueChunk lds[kPiece * kEntryChunks<R>];
#ifdef RTX_BOUNCE_LDS_PAD  // occupancy experiments: LDS nobody uses limits the workgroups per CU
    __shared__ uint32_t lds_pad[RTX_BOUNCE_LDS_PAD / 4];
    if (threadIdx.x == 0xffffffu) lds_pad[blockIdx.x] = 1u;
    asm volatile("" :: "v"(lds_pad[threadIdx.x]));
#endif
    const RT_CONSTANT_AS BounceArgs<R> *kp =
        (const RT_CONSTANT_AS BounceArgs<R> *)__builtin_amdgcn_kernarg_segment_ptr();
    const uint32_t sub_in = blockIdx.y;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t step = gridDim.x * kPiece;
    // one ray per wave for short levels: scenes with the sphere hierarchy whose rays carry no container
    // bookkeeping (that is kept per lane and not joined across lanes)
    constexpr bool kWide = (SF & SF_CULL) != 0 && (SF & SF_TRANSPARENT) == 0 && !kLaneWalk;
    constexpr uint32_t kWideRays = RTX_WIDE_RAYS;
#ifdef RTX_LDS_TABLES
    if constexpr ((SF & SF_CULL) == 0) stage_tables<R>(args_in_kernarg_segment.S);
#endif
    if constexpr (COLD_LDS) stage_cold_table<R>(args_in_kernarg_segment.S, kPiece);
    V3<R> ro = { R(0), R(0), R(0) }, rd = ro;
    R weight = R(0);
    uint32_t pixel = 0;
    bool first = true;
    RTX_PHASE_INIT();
#ifdef RTX_PHASE_CLOCK
    // one wave per launch compares the shader clock with the constant 100 MHz clock: the ratio
    // is the frequency the chip ran at while this launch was resident
    const unsigned long long clk0 = __builtin_readcyclecounter(), real0 = wall_clock64();
#endif
    for (uint32_t base = blockIdx.x * kPiece;; base += step) {
        // the compiler must not know that kp is the same pointer as in the last round, or it
        // hoists every load from it out of the loop (some fifty scalar registers)
        asm volatile("" : "+s"(kp));
        BounceArgs<R> A;
        __builtin_memcpy(&A, kp, sizeof A);
        uint32_t n = ro_load(A.in_q.count + (size_t)sub_in * kCountStride);
        if (n > A.in_q.cap_sub) n = A.in_q.cap_sub;
        // With bounding-sphere tests a wave walks the union of the spheres its rays can hit, one
        // dependent fetch after another, and deep levels hold few, incoherent rays: their waves
        // are long chains on a mostly idle chip.  While a level is short enough that the whole
        // chip has lanes to spare, give every ray `spread` lanes' worth of room (one ray per
        // `spread` lanes): more, shorter walks side by side.  n * spread stays within the
        // sub-queue's capacity rounded up to whole blocks, so the block indices, and with them
        // the bound on the output sub-queues, do not grow.
        uint32_t spread = 1u;
        if constexpr ((SF & SF_CULL) != 0 && !kLaneWalk) {
            const uint32_t budget = kSpreadLanes / gridDim.y;  // lanes per sub-queue when the chip is lightly loaded
            const uint64_t room = ((uint64_t)A.in_q.cap_sub + kPiece - 1) / kPiece * kPiece;
            while (spread < 64u && (uint64_t)n * (spread * 2u) <= budget && (uint64_t)n * (spread * 2u) <= room) spread *= 2u;
            // A level of at most kWideRays rays gets a wave per ray, whose 64 lanes walk the sphere hierarchy
            // side by side (wide_walk_table): 64 waves per SIMD at most, each some 30 steps long.
            if constexpr (kWide) {
                if ((uint64_t)n * gridDim.y <= kWideRays && (uint64_t)n * 64u <= room) spread = 64u;
            }
        }
        const bool wide = kWide && spread == 64u;  // wave-uniform
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
            entry_prefetch<R>(lds, wave, valid, A.in_q, sub_in, k);
            entry_take<R>(lds, wave, lane, valid, ro, rd, weight, pixel);
            first = false;
        }
        const bool more = base + step < total;  // wave-uniform
        const bool valid_next = more && slot_of(base + step, k_next);
        entry_prefetch<R>(lds, wave, valid_next, A.in_q, sub_in, k_next);

#ifdef RTX_PHASE_CLOCK
        RTX_PHASE_BEGIN();
#endif
        Node<R> nd;
        nd.hit = false;
        nd.has_reflect = nd.has_refract = false;
        nd.lighting = { R(0), R(0), R(0) };
        const uint32_t block = base / kPiece;
        const uint32_t sub = (block + sub_in) & A.out_q.sub_mask;
        const R w = weight;
        const uint32_t px_index = pixel;
#ifdef RTX_PHASE_CLOCK
        const V3<R> dbg_o = ro, dbg_d = rd;
#endif
        bool done = false;
        if constexpr (kWide) {
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
                    nd = trace_node<R, SF, false, WALK_WIDE, COLD_LDS>(A.S, nullptr, o, d, A.remaining, [&](const Node<R> &c) {
                        queue_push2(A.out_q, sub, c.has_reflect && lane == 0u, c.over_point, c.reflectv, w * c.reflect_factor,
                                    c.has_refract && lane == 0u, c.under_point, c.refract_dir, w * c.refract_factor, px_index);
                    });
                }
                done = true;
            }
        }
        if (!done && valid) {
            nd = trace_node<R, SF, false, (kLaneWalk && (SF & SF_CULL) != 0) ? WALK_LANE : WALK_WAVE, COLD_LDS>(A.S, nullptr, ro, rd, A.remaining, [&](const Node<R> &c) {
                queue_push2(A.out_q, sub, c.has_reflect, c.over_point, c.reflectv, w * c.reflect_factor, c.has_refract,
                            c.under_point, c.refract_dir, w * c.refract_factor, px_index);
            });
        }
        count_hits(A.stats, block + sub_in, A.level, valid && nd.hit);
        entry_take<R>(lds, wave, lane, valid_next, ro, rd, weight, pixel);
#ifdef RTX_EXP_NO_FB_ADD  // knock-out experiment: the frame is wrong, the time shows what the atomics cost
        if (valid && nd.hit && nd.lighting.x * w == R(-12345)) {
#else
        if (valid && nd.hit) {
#endif
            R *px = A.fb + (size_t)px_index * 3;
            fb_add<R>(px + 0, nd.lighting.x * w);
            fb_add<R>(px + 1, nd.lighting.y * w);
            fb_add<R>(px + 2, nd.lighting.z * w);
        }
#ifdef RTX_PHASE_CLOCK
        {   // the longest round of any wave (a whole color_at node for 64 rays), and how many took over 100 000 cycles
            const unsigned long long took = __builtin_readcyclecounter() - phase_t_;
            if (phase_leader()) {
                atomicMax(&g_phase[13], took);
                if (took > 100000ull) atomicAdd(&g_phase[12], 1ull);
            }
            // the rays of the first wave whose round took over g_slow_threshold cycles
            unsigned int claimed = 1u;
            if (took > g_slow_threshold && phase_leader()) claimed = atomicCAS(&g_slow_flag, 0u, 1u);
            claimed = __shfl(claimed, (int)(__ffsll((long long)__ballot(1)) - 1));
            if (took > g_slow_threshold && claimed == 0u) {
                double *o = g_slow_rays + (threadIdx.x & 63u) * 8;
                o[0] = dbg_o.x; o[1] = dbg_o.y; o[2] = dbg_o.z; o[3] = dbg_d.x; o[4] = dbg_d.y; o[5] = dbg_d.z;
                o[6] = valid ? 1.0 : 0.0; o[7] = (double)took;
            }
        }
        RTX_PHASE_MARK(5);  // a whole round, for the waves that reach here converged
#endif
        if (!more) break;
    }
#ifdef RTX_PHASE_CLOCK
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        atomicAdd(&g_phase[14], __builtin_readcyclecounter() - clk0);
        atomicAdd(&g_phase[15], wall_clock64() - real0);
    }
#endif
    RTX_PHASE_FLUSH();
}

#ifdef RTX_SPLIT_BOUNCE
// Experiment (tools build, -DRTX_SPLIT_BOUNCE; measured on a 4K frame of an 8-object scene with glass: 208 + 314 us
// per level for the two kernels against 371 us for the fused one, the same vector instructions at twice the occupancy):
// one recursion level as two launches with the hit record in memory between them -- the
// intersect kernel finds every queued ray's closest hit, the shade kernel does the rest of the
// color_at node.  One ray per thread, grids that cover the queues' capacity.
template <typename R>
struct alignas(16) HitRec {
    R t;
    uint32_t slot, obj, top1, top2, open, pad;
};
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
template <typename R, uint32_t SF>
__global__ void __launch_bounds__(bounce_block<R>(SF)) k_split_intersect(BounceArgs<R> A, HitRec<R> *__restrict__ hits)
{
    const uint32_t sub = blockIdx.y;
    uint32_t n = ro_load(A.in_q.count + (size_t)sub * kCountStride);
    if (n > A.in_q.cap_sub) n = A.in_q.cap_sub;
    if (blockIdx.x * bounce_block<R>(SF) >= n) return;
    const uint32_t k = blockIdx.x * bounce_block<R>(SF) + threadIdx.x;
    if (k >= n) return;
    V3<R> ro, rd;
    R weight;
    uint32_t pixel;
    entry_load(A.in_q, sub, k, ro, rd, weight, pixel);
    const Hit<R> h = closest_hit<R, (SF & SF_TRANSPARENT) != 0, (SF & SF_CULL) != 0, (SF & SF_CHAIN) != 0>(A.S, ro, rd, Num<R>::inf);
    HitRec<R> out = { h.t, h.slot, h.obj, h.top1, h.top2, h.hit_open ? 1u : 0u, 0u };
    hits[(size_t)sub * A.in_q.cap_sub + k] = out;
}
template <typename R, uint32_t SF>
__global__ void __launch_bounds__(bounce_block<R>(SF)) k_split_shade(BounceArgs<R> A, const HitRec<R> *__restrict__ hits)
{
    const uint32_t sub_in = blockIdx.y;
    uint32_t n = ro_load(A.in_q.count + (size_t)sub_in * kCountStride);
    if (n > A.in_q.cap_sub) n = A.in_q.cap_sub;
    if (blockIdx.x * bounce_block<R>(SF) >= n) return;
    const uint32_t k = blockIdx.x * bounce_block<R>(SF) + threadIdx.x;
    const bool valid = k < n;
    Node<R> nd;
    nd.hit = false;
    nd.has_reflect = nd.has_refract = false;
    nd.lighting = { R(0), R(0), R(0) };
    R weight = R(0);
    uint32_t pixel = 0;
    const uint32_t sub = (blockIdx.x + sub_in) & A.out_q.sub_mask;
    if (valid) {
        V3<R> ro, rd;
        entry_load(A.in_q, sub_in, k, ro, rd, weight, pixel);
        const HitRec<R> r = hits[(size_t)sub_in * A.in_q.cap_sub + k];
        Hit<R> h;
        h.t = r.t; h.slot = r.slot; h.obj = r.obj; h.top1 = r.top1; h.top2 = r.top2; h.hit_open = r.open != 0u;
        h.top1_t = h.top2_t = R(0);
        nd = shade_node<R, SF, false, WALK_WAVE, false>(A.S, ro, rd, A.remaining, h, [&](const Node<R> &c) {
            queue_push2(A.out_q, sub, c.has_reflect, c.over_point, c.reflectv, weight * c.reflect_factor, c.has_refract, c.under_point,
                        c.refract_dir, weight * c.refract_factor, pixel);
        });
        if (nd.hit) {
            R *px = A.fb + (size_t)pixel * 3;
            fb_add<R>(px + 0, nd.lighting.x * weight);
            fb_add<R>(px + 1, nd.lighting.y * weight);
            fb_add<R>(px + 2, nd.lighting.z * weight);
        }
    }
    count_hits(A.stats, blockIdx.x + sub_in, A.level, valid && nd.hit);
}
#endif

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
#ifdef RTX_SPLIT_BOUNCE
    {
        static void *hit_buffer = nullptr;  // experiment: one buffer for the process, never shrunk (single-threaded use)
        static size_t hit_bytes = 0;
        const size_t need = (size_t)n_sub * in_q.cap_sub * sizeof(HitRec<R>);
        if (need > hit_bytes) {
            (void)hipDeviceSynchronize();
            if (hit_buffer) (void)hipFree(hit_buffer);
            hit_buffer = nullptr;
            hit_bytes = 0;
            const hipError_t e = hipMalloc(&hit_buffer, need);
            if (e != hipSuccess) return e;
            hit_bytes = need;
        }
        const dim3 full(worst, n_sub);
        HitRec<R> *hits = static_cast<HitRec<R> *>(hit_buffer);
#define RTX_CALL(SFV)                                                                                   \
    if constexpr (((SFV) & (SF_REFLECTIVE | SF_TRANSPARENT)) != 0) {                                    \
        hipLaunchKernelGGL((k_split_intersect<R, SFV | SF_PATTERN>), full, dim3(piece), 0, stream, args, hits); \
        hipLaunchKernelGGL((k_split_shade<R, SFV | SF_PATTERN>), full, dim3(piece), 0, stream, args, hits);     \
    }
        RTX_SF_SWITCH(sf, RTX_CALL)
#undef RTX_CALL
        return hipGetLastError();
    }
#endif
    // Scenes small enough keep their shading table in LDS; only the variants without bounding-sphere
    // tests exist in that form (a scene with those tests has at least 16 objects and usually thousands).
    const uint32_t cold_bytes =
#ifdef RTX_LDS_TABLES
        0u;  // that experiment uses the dynamic LDS for the intersection tables
#else
        (sf & SF_CULL) ? 0u : cold_lds_bytes<R>(scene.n);
#endif
#define RTX_CALL(SFV)                                                                                   \
    if constexpr (((SFV) & (SF_REFLECTIVE | SF_TRANSPARENT)) != 0) {                                    \
        if constexpr (((SFV) & SF_CULL) == 0) {                                                         \
            if (cold_bytes)                                                                             \
                hipLaunchKernelGGL((k_bounce<R, SFV | SF_PATTERN, true>), grid, dim3(piece), cold_bytes, stream, args); \
            else                                                                                        \
                hipLaunchKernelGGL((k_bounce<R, SFV | SF_PATTERN, false>), grid, dim3(piece), table_lds_bytes(scene), stream, args); \
        } else {                                                                                        \
            hipLaunchKernelGGL((k_bounce<R, SFV | SF_PATTERN, false>), grid, dim3(piece), table_lds_bytes(scene), stream, args); \
        }                                                                                               \
    }
    RTX_SF_SWITCH(sf, RTX_CALL)
#undef RTX_CALL
    return hipGetLastError();
}

}  // namespace rtx
