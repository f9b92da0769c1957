// render_f32.hip -- the f32 fast path: the same kernels instantiated for float, built with
// floating-point contraction on (FMA) since bit parity is not the goal here; its colours
// are held to 1/255 after 8-bit quantisation against the f64 result.
#include "rt_kernels.h"

namespace rtx {
template struct Launch<float>;
}  // namespace rtx

