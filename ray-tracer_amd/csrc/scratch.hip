// scratch.hip -- device memory for the render call that allocates nothing
// (rtx_render_device_noalloc in rtx_api.cpp): one file-scope array per GPU, present from the
// moment the library's code object is loaded on a device until the process ends.
#include <hip/hip_runtime.h>

#include "rt_launch.h"

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

}  // namespace rtx
