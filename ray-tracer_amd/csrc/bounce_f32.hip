// bounce_f32.hip -- the secondary-ray kernel in f32 (see rt_bounce.h for why it is a unit of its own).
#include "rt_bounce.h"

namespace rtx {
template hipError_t launch_bounce<float>(uint32_t, hipStream_t, const SceneDev<float> &, const QueueDev<float> &, const QueueDev<float> &,
                                       uint32_t, uint32_t, float *, StatShard *, uint32_t, uint32_t);
}  // namespace rtx

#ifdef RTX_PHASE_CLOCK
// diagnostic build: the rays of the first bounce wave whose round took more than `threshold` cycles
// since the last call (64 x {origin, direction, valid, cycles}); arms the capture again
extern "C" int rtx_debug_slow_rays_f32(double *out, unsigned long long threshold)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rtx::g_slow_rays), sizeof(double) * 64 * 8) != hipSuccess) return -1;
    const unsigned int zero = 0u;
    if (hipMemcpyToSymbol(HIP_SYMBOL(rtx::g_slow_flag), &zero, sizeof zero) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(rtx::g_slow_threshold), &threshold, sizeof threshold) != hipSuccess) return -1;
    return 0;
}
extern "C" int rtx_debug_phase_bounce_f32(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rtx::g_phase), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(rtx::g_phase), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif
