// render_f32.hip -- the f32 fast path: the same kernels instantiated for float, built with
// floating-point contraction on (FMA) since bit parity is not the goal here; its colours
// are held to 1/255 after 8-bit quantisation against the f64 result.
#include "rt_kernels.h"

namespace rtx {
template struct Launch<float>;
}  // namespace rtx

#ifdef RTX_COUNT_WALK
extern "C" int rtx_debug_walk_f32(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rtx::g_walk), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(rtx::g_walk), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif
