// bounce_f64.hip -- the secondary-ray kernel in f64: built with -ffp-contract=off like render_f64.hip (see rt_bounce.h for why it is a unit of its own).
#include "rt_bounce.h"

#if defined(__FP_FAST_FMA) || defined(__FAST_MATH__)
#error "bounce_f64.hip must be compiled with -ffp-contract=off and without fast-math"
#endif

namespace rtx {
template hipError_t launch_bounce<double>(uint32_t, hipStream_t, const SceneDev<double> &, const QueueDev<double> &, const QueueDev<double> &,
                                       uint32_t, uint32_t, double *, StatShard *, uint32_t, uint32_t);
}  // namespace rtx

#ifdef RTX_PHASE_CLOCK
// diagnostic build: the rays of the first bounce wave whose round took more than `threshold` cycles
// since the last call (64 x {origin, direction, valid, cycles}); arms the capture again
extern "C" int rtx_debug_slow_rays_f64(double *out, unsigned long long threshold)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rtx::g_slow_rays), sizeof(double) * 64 * 8) != hipSuccess) return -1;
    const unsigned int zero = 0u;
    if (hipMemcpyToSymbol(HIP_SYMBOL(rtx::g_slow_flag), &zero, sizeof zero) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(rtx::g_slow_threshold), &threshold, sizeof threshold) != hipSuccess) return -1;
    return 0;
}
extern "C" int rtx_debug_phase_bounce(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rtx::g_phase), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(rtx::g_phase), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif
