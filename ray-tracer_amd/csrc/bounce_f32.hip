// bounce_f32.hip -- the secondary-ray kernel in f32 (see rt_bounce.h for why it is a unit of its own).
#include "rt_bounce.h"

namespace rtx {
template hipError_t launch_bounce<float>(uint32_t, hipStream_t, const SceneDev<float> &, const QueueDev<float> &, const QueueDev<float> &,
                                       uint32_t, uint32_t, float *, StatShard *, uint32_t, uint32_t);
}  // namespace rtx

