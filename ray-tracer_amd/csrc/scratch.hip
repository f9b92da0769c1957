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

}  // namespace rtx
