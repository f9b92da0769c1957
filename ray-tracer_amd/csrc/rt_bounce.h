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
// row read its actual length on the device and share its rays, kRayBlock at a time (workgroup x
// takes blocks x, x + gridDim.x, ...): the launch has as many waves as the chip holds, whatever
// the worst-case capacity of the queue is.  The children of block b of sub-queue y go to output
// sub-queue (b + y) mod n_sub, so every output sub-queue is fed by at most one block per block
// index: ceil(cap_sub / kRayBlock) * kRayBlock * children-per-ray entries, which is the bound
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
template <typename R, uint32_t SF>
__global__ void __launch_bounds__(kRayBlock<R>)
k_bounce(BounceArgs<R> args_in_kernarg_segment)
{
    __shared__ QueueChunk lds[kRayBlock<R> * kEntryChunks<R>];
#ifdef RTX_BOUNCE_LDS_PAD  // occupancy experiments: LDS nobody uses limits the workgroups per CU
    __shared__ uint32_t lds_pad[RTX_BOUNCE_LDS_PAD / 4];
    if (threadIdx.x == 0xffffffu) lds_pad[blockIdx.x] = 1u;
    asm volatile("" :: "v"(lds_pad[threadIdx.x]));
#endif
    const RT_CONSTANT_AS BounceArgs<R> *kp =
        (const RT_CONSTANT_AS BounceArgs<R> *)__builtin_amdgcn_kernarg_segment_ptr();
    const uint32_t sub_in = blockIdx.y;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t step = gridDim.x * kRayBlock<R>;
#ifdef RTX_LDS_TABLES
    if constexpr ((SF & SF_CULL) == 0) stage_tables<R>(args_in_kernarg_segment.S);
#endif
    V3<R> ro = { R(0), R(0), R(0) }, rd = ro;
    R weight = R(0);
    uint32_t pixel = 0;
    bool first = true;
    RTX_PHASE_INIT();
    for (uint32_t base = blockIdx.x * kRayBlock<R>;; base += step) {
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
            const uint64_t room = ((uint64_t)A.in_q.cap_sub + kRayBlock<R> - 1) / kRayBlock<R> * kRayBlock<R>;
            while (spread < 64u && (uint64_t)n * (spread * 2u) <= budget && (uint64_t)n * (spread * 2u) <= room) spread *= 2u;
        }
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
        const uint32_t block = base / kRayBlock<R>;
        const uint32_t sub = (block + sub_in) & A.out_q.sub_mask;
        const R w = weight;
        const uint32_t px_index = pixel;
        if (valid) {
            nd = trace_node<R, SF, false, kLaneWalk && (SF & SF_CULL) != 0>(A.S, nullptr, ro, rd, A.remaining, [&](const Node<R> &c) {
                queue_push2(A.out_q, sub, c.has_reflect, c.over_point, c.reflectv, w * c.reflect_factor, c.has_refract,
                            c.under_point, c.refract_dir, w * c.refract_factor, px_index);
            });
        }
        count_hits(A.stats, block + sub_in, A.level, valid && nd.hit);
        entry_take<R>(lds, wave, lane, valid_next, ro, rd, weight, pixel);
        if (valid && nd.hit) {
            R *px = A.fb + (size_t)px_index * 3;
            fb_add<R>(px + 0, nd.lighting.x * w);
            fb_add<R>(px + 1, nd.lighting.y * w);
            fb_add<R>(px + 2, nd.lighting.z * w);
        }
#ifdef RTX_PHASE_CLOCK
        RTX_PHASE_MARK(5);  // a whole round, for the waves that reach here converged
#endif
        if (!more) break;
    }
    RTX_PHASE_FLUSH();
}

template <typename R>
hipError_t launch_bounce(uint32_t sf, hipStream_t stream, const SceneDev<R> &scene, const QueueDev<R> &in_q,
                             const QueueDev<R> &out_q, uint32_t remaining, uint32_t level, R *fb, StatShard *stats,
                             uint32_t resident_blocks)
{
    if ((sf & (SF_REFLECTIVE | SF_TRANSPARENT)) == 0 || in_q.cap_sub == 0) return hipSuccess;  // nothing can be queued
    const uint32_t n_sub = in_q.sub_mask + 1;
    const uint32_t worst = (in_q.cap_sub + kRayBlock<R> - 1) / kRayBlock<R>;
    const uint32_t share = resident_blocks / n_sub > 0 ? resident_blocks / n_sub : 1u;
    const dim3 grid(worst < share ? worst : share, n_sub);
    const BounceArgs<R> args = { scene, in_q, out_q, remaining, level, fb, stats };
#define RTX_CALL(SFV)                                                                                   \
    if constexpr (((SFV) & (SF_REFLECTIVE | SF_TRANSPARENT)) != 0)                                      \
        hipLaunchKernelGGL((k_bounce<R, SFV | SF_PATTERN>), grid, dim3(kRayBlock<R>), table_lds_bytes(scene), stream, args)
    RTX_SF_SWITCH(sf, RTX_CALL)
#undef RTX_CALL
    return hipGetLastError();
}

}  // namespace rtx
