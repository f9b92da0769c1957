// scratch.hip -- device memory for the render call that allocates nothing
// (rtx_render_device_noalloc in rtx_api.cpp): one file-scope array per GPU, present from the
// moment the library's code object is loaded on a device until the process ends.
#include <hip/hip_runtime.h>

#include "rt_launch.h"
#include "rt_types.h"

namespace rtx {

// Holds, for one frame at a time: the hit counters, the queue length counters of every row
// chunk and recursion level, the per-frame primary-ray tables and both ping-pong ray queues.
static __device__ __align__(256) __attribute__((used)) unsigned char scratch[kScratchBytes];

// Device address of the array on the CURRENT device.  `scratch` itself is only a host-side handle
// for the symbol: the address a kernel can use comes from the runtime, separately for every GPU
// (each holds its own copy of the array at its own address).
hipError_t scratch_address(void **ptr, size_t *bytes)
{
    *ptr = nullptr;
    *bytes = 0;
    const hipError_t e = hipGetSymbolAddress(ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    *bytes = kScratchBytes;
    return hipSuccess;
}

// The lengths of one level's sub-queues, written where the host can read them without a copy of
// its own: `host` is pinned host memory mapped into the device (out[1 + q] = length of sub-queue q,
// then out[0] = seq, which the host polls for).  One workgroup; counters lie `stride` words apart.
__global__ void __launch_bounds__(256)
k_publish_counts(const uint32_t *__restrict__ counts, uint32_t n_sub, uint32_t stride, uint32_t *__restrict__ host, uint32_t seq)
{
    for (uint32_t q = threadIdx.x; q < n_sub; q += 256u)
        __hip_atomic_store(host + 1 + q, counts[(size_t)q * stride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_publish_counts(hipStream_t stream, const uint32_t *counts, uint32_t n_sub, uint32_t stride, uint32_t *host, uint32_t seq)
{
    hipLaunchKernelGGL(k_publish_counts, dim3(1), dim3(256), 0, stream, counts, n_sub, stride, host, seq);
    return hipGetLastError();
}


// The block list of a regional queue: {sub-queue, block} for every block of kListBlock entries that holds a ray,
// sub-queue by sub-queue, and their number.  Every workgroup scans the lengths of the n_sub <= kListMaxSub
// sub-queues for itself (a thread takes a run of consecutive sub-queues, whole regions) and then writes its share
// of the list.
// next_lines != null: the queue of the level after this one will be sized from these lengths (a call that reads
// them back on the host and allocates accordingly): each of the `kinds` sub-queues of a screen region gets room
// for as many entries as the region's sub-queues hold now (a ray has at most one child of a kind), back to back;
// where each begins and what it holds goes into the counter lines of that next queue (kLineBegin, kLineCap).
constexpr uint32_t kListThreads = 256;
__global__ void __launch_bounds__(kListThreads)
k_list_blocks(const uint32_t *__restrict__ count, uint32_t n_sub, uint32_t cap_sub, uint32_t var, QueueDev<float>::Block *__restrict__ list,
              uint32_t *__restrict__ n_list, uint32_t *__restrict__ next_lines, uint32_t kinds)
{
    __shared__ uint32_t before[kListMaxSub + 1];  // before[q]: blocks of the sub-queues 0..q-1
    __shared__ uint32_t of_wave[kListThreads / 64u], rays_of_wave[kListThreads / 64u];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t per = (n_sub + kListThreads - 1u) / kListThreads;
    per = (per + kinds - 1u) / kinds * kinds;
    const uint32_t q0 = threadIdx.x * per;
    auto length = [&](uint32_t q) {
        uint32_t n = count[(size_t)q * kCountStride];
        const uint32_t cap = var ? count[(size_t)q * kCountStride + kLineCap] : cap_sub;
        return n > cap ? cap : n;
    };
    uint32_t mine = 0u, my_rays = 0u;
    for (uint32_t q = q0; q < q0 + per && q < n_sub; ++q) {
        const uint32_t n = length(q);
        my_rays += n;
        mine += (n + kListBlock - 1u) / kListBlock;
        before[q + 1u] = mine;  // within this thread's run for now
    }
    uint32_t incl = mine, incl_rays = my_rays;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off), up_rays = (uint32_t)__shfl_up((int)incl_rays, off);
        if ((int)lane >= off) { incl += up; incl_rays += up_rays; }
    }
    if (lane == 63u) { of_wave[wave] = incl; rays_of_wave[wave] = incl_rays; }
    __syncthreads();
    for (uint32_t w = 0; w < wave; ++w) { incl += of_wave[w]; incl_rays += rays_of_wave[w]; }
    const uint32_t ahead = incl - mine;  // blocks of the sub-queues before this thread's run
    for (uint32_t q = q0; q < q0 + per && q < n_sub; ++q) before[q + 1u] += ahead;
    if (threadIdx.x == 0u) before[0] = 0u;
    if (next_lines && blockIdx.x == 0u) {
        uint32_t begin = (incl_rays - my_rays) * kinds;  // entries of the regions before this thread's run
        for (uint32_t q = q0; q < q0 + per && q < n_sub; q += kinds) {
            uint32_t region = 0u;
            for (uint32_t k = 0; k < kinds && q + k < n_sub; ++k) region += length(q + k);
            for (uint32_t k = 0; k < kinds && q + k < n_sub; ++k) {
                next_lines[(size_t)(q + k) * kCountStride + kLineBegin] = begin + k * region;
                next_lines[(size_t)(q + k) * kCountStride + kLineCap] = region;
            }
            begin += kinds * region;
        }
    }
    __syncthreads();
    const uint32_t total = before[n_sub];
    if (blockIdx.x == 0u && threadIdx.x == 0u) *n_list = total;
    for (uint32_t i = blockIdx.x * kListThreads + threadIdx.x; i < total; i += gridDim.x * kListThreads) {
        uint32_t lo = 0u, hi = n_sub;  // the last q with before[q] <= i
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (before[mid] <= i) lo = mid; else hi = mid;
        }
        list[i] = QueueDev<float>::Block{ lo, i - before[lo] };
    }
}
hipError_t list_blocks_launch(hipStream_t stream, const uint32_t *count, uint32_t n_sub, uint32_t cap_sub, uint32_t var, void *list,
                              uint32_t *n_list, uint32_t *next_lines, uint32_t kinds)
{
    if (n_sub > kListMaxSub || kinds == 0u || n_sub % kinds != 0u) return hipErrorInvalidValue;
    // enough workgroups that a thread writes a handful of entries in a 4K frame (some 20 000 blocks per level)
    hipLaunchKernelGGL(k_list_blocks, dim3(32), dim3(kListThreads), 0, stream, count, n_sub, cap_sub, var,
                       static_cast<QueueDev<float>::Block *>(list), n_list, next_lines, kinds);
    return hipGetLastError();
}

}  // namespace rtx
