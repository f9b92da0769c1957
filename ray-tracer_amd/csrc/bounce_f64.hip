// bounce_f64.hip -- the secondary-ray kernel in f64: built with -ffp-contract=off like render_f64.hip (see rt_bounce.h for why it is a unit of its own).
#include "rt_bounce.h"

#if defined(__FP_FAST_FMA) || defined(__FAST_MATH__)
#error "bounce_f64.hip must be compiled with -ffp-contract=off and without fast-math"
#endif

namespace rtx {
template hipError_t launch_bounce<double>(uint32_t, hipStream_t, const SceneDev<double> &, const QueueDev<double> &, const QueueDev<double> &,
                                       uint32_t, uint32_t, double *, StatShard *, uint32_t, uint32_t);
}  // namespace rtx

