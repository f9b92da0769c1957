// render_f64.hip -- the f64 parity path.  Build with -ffp-contract=off: the reference is
// Rust, whose f64 arithmetic never fuses a multiply with an add, and one differing ulp in a
// hit distance can flip a pattern cell, a shadow test or a `t >= 0` decision.
#include "rt_kernels.h"

#if defined(__FP_FAST_FMA) || defined(__FAST_MATH__)
#error "render_f64.hip must be compiled with -ffp-contract=off and without fast-math"
#endif

namespace rtx {

// One color_at node per ray with every intermediate kept: the parity tests compare these
// with intersect_world / prepare_computations / is_shadowed / lighting of the oracle.
template <uint32_t SF>
__global__ void __launch_bounds__(kBlock)
k_probe(SceneDev<double> S, const double *__restrict__ rays, uint32_t n, rtx_probe *__restrict__ out)
{
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n; k += stride) {
        const double *r = rays + (size_t)k * 6;
        V3<double> ro = { r[0], r[1], r[2] }, rd = { r[3], r[4], r[5] };
        Node<double> nd = trace_node<double, SF, false, WALK_WAVE, false, true>(S, nullptr, ro, rd, 5u);
        rtx_probe p;
        __builtin_memset(&p, 0, sizeof p);
        p.t = -1.0;
        p.object = -1;
        if (nd.hit) {
            p.t = nd.t;
            p.object = (int32_t)nd.obj;
            p.point[0] = nd.point.x; p.point[1] = nd.point.y; p.point[2] = nd.point.z;
            p.normalv[0] = nd.normalv.x; p.normalv[1] = nd.normalv.y; p.normalv[2] = nd.normalv.z;
            p.over_point[0] = nd.over_point.x; p.over_point[1] = nd.over_point.y; p.over_point[2] = nd.over_point.z;
            p.under_point[0] = nd.under_point.x; p.under_point[1] = nd.under_point.y; p.under_point[2] = nd.under_point.z;
            p.reflectv[0] = nd.reflectv.x; p.reflectv[1] = nd.reflectv.y; p.reflectv[2] = nd.reflectv.z;
            p.n1 = nd.n1; p.n2 = nd.n2;
            p.lighting[0] = nd.lighting.x; p.lighting[1] = nd.lighting.y; p.lighting[2] = nd.lighting.z;
            p.inside = nd.inside ? 1 : 0;
            p.shadowed = nd.shadowed ? 1 : 0;
        }
        out[k] = p;
    }
}

hipError_t launch_probe(uint32_t sf, uint32_t grid, hipStream_t stream, const SceneDev<double> &scene,
                        const double *rays, uint32_t n, rtx_probe *out)
{
#define RTX_CALL(SFV) \
    hipLaunchKernelGGL((k_probe<SFV | SF_PATTERN>), dim3(grid), dim3(kBlock), 0, stream, scene, rays, n, out)
    RTX_SF_SWITCH(sf, RTX_CALL)
#undef RTX_CALL
    return hipGetLastError();
}

// Self-check of div3, the square root and the fused normalisation against the compiler's plain
// divisions and roots on n pseudo-random operand sets: exponents over the whole f64 range (so
// both branches run), zeros, infinities, NaNs and denormals among them.
// Counts the sets where any quotient differs in its bits (NaNs compare as equal to NaNs).
__global__ void __launch_bounds__(kBlock)
k_check_div3(uint64_t seed, uint32_t n, uint32_t narrow, unsigned long long *mismatches)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    uint64_t st = seed + (uint64_t)k * 0x9e3779b97f4a7c15ull;
    auto next = [&]() {
        st += 0x9e3779b97f4a7c15ull;
        uint64_t z = st;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    };
    auto operand = [&]() {
        uint64_t bits = next();
        if (narrow) {  // exponent near 0: the shared-reciprocal branch
            const uint64_t e = 1023u - 40u + (next() % 80u);
            bits = (bits & 0x800fffffffffffffull) | (e << 52);
        } else if ((next() & 15u) == 0u) {
            const uint64_t special[6] = { 0ull, 0x8000000000000000ull, 0x7ff0000000000000ull, 0xfff0000000000000ull,
                                          0x7ff8000000000000ull, 0x0000000000000001ull };
            bits = special[next() % 6u];
        }
        return __builtin_bit_cast(double, bits);
    };
    const V3<double> a = { operand(), operand(), operand() };
    const double b = operand();
    const V3<double> q = div3(a, b);
    const double want[3] = { a.x / b, a.y / b, a.z / b }, got[3] = { q.x, q.y, q.z };
    bool bad = false;
    for (int i = 0; i < 3; ++i) {
        const bool both_nan = want[i] != want[i] && got[i] != got[i];
        bad = bad || (!both_nan && __builtin_bit_cast(uint64_t, want[i]) != __builtin_bit_cast(uint64_t, got[i]));
    }
    // the square root with and without its range handling, and the fused a / |a|
    const double x = __builtin_fabs(b);
    const double root = Num<double>::sqrt_(x), root_want = __builtin_sqrt(x);
    bad = bad || (!(root != root && root_want != root_want) && __builtin_bit_cast(uint64_t, root) != __builtin_bit_cast(uint64_t, root_want));
    const double neg_root = Num<double>::sqrt_(-x);  // NaN, or -0 for -0
    bad = bad || (x != 0.0 && neg_root == neg_root);
    const double m2 = (a.x * a.x + a.y * a.y) + a.z * a.z;
    const V3<double> u = normalized(a, m2);
    const double mag = __builtin_sqrt(m2);
    const double u_want[3] = { a.x / mag, a.y / mag, a.z / mag }, u_got[3] = { u.x, u.y, u.z };
    for (int i = 0; i < 3; ++i) {
        const bool both_nan = u_want[i] != u_want[i] && u_got[i] != u_got[i];
        bad = bad || (!both_nan && __builtin_bit_cast(uint64_t, u_want[i]) != __builtin_bit_cast(uint64_t, u_got[i]));
    }
    if (bad) atomicAdd(mismatches, 1ull);
}

hipError_t launch_check_div3(hipStream_t stream, uint64_t seed, uint32_t n, uint32_t narrow, unsigned long long *mismatches)
{
    hipLaunchKernelGGL(k_check_div3, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, seed, n, narrow, mismatches);
    return hipGetLastError();
}

// Self-check of pow_specular against the general-purpose pow on n pseudo-random (x, y): x in
// (0, 1] uniformly, or within 1e-3 of 1 (where specular highlights live), y among the common
// shininess values and uniformly in [0, 2000].  out[0]: largest absolute difference, out[1]:
// largest difference relative to the result (results above 1e-300), as the bits of non-negative
// doubles (which order like integers).
__global__ void __launch_bounds__(kBlock)
k_check_pow(uint64_t seed, uint32_t n, unsigned long long *out)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    uint64_t st = seed + (uint64_t)k * 0x9e3779b97f4a7c15ull;
    auto next = [&]() {
        st += 0x9e3779b97f4a7c15ull;
        uint64_t z = st;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    };
    auto unit = [&]() { return (double)((next() >> 11) + 1ull) * (1.0 / 9007199254740992.0); };  // (0, 1]
    const double x = (next() % 3u) ? unit() : 1.0 - (unit() - 1.0 / 9007199254740992.0) * 1e-3;
    const double common[5] = { 10.0, 50.0, 200.0, 300.0, 1000.0 };
    const uint64_t pick = next() % 6u;
    const double y = pick < 5u ? common[pick] : unit() * 2000.0;
    const double got = pow_specular(x, y), want = pow(x, y);
    const double diff = __builtin_fabs(got - want);
    atomicMax(out + 0, __builtin_bit_cast(unsigned long long, diff));
    if (want > 1e-300) atomicMax(out + 1, __builtin_bit_cast(unsigned long long, diff / want));
    // the special cases: NaN in, NaN out; y = 0; a power that overflows; one that underflows
    if (k == 0) {
        const double nan = __builtin_nan("");
        const double a = pow_specular(nan, 200.0), b = pow_specular(0.5, nan), c = pow_specular(0.3, 0.0),
                     d = pow_specular(0.5, -5000.0), e = pow_specular(0.5, 5000.0), f = pow_specular(1.0, 1e300);
        const bool ok = a != a && b != b && c == 1.0 && d == __builtin_huge_val() && e == 0.0 && f == 1.0;
        if (!ok) atomicMax(out + 0, __builtin_bit_cast(unsigned long long, 1.0));
    }
}

hipError_t launch_check_pow(hipStream_t stream, uint64_t seed, uint32_t n, unsigned long long *out)
{
    hipLaunchKernelGGL(k_check_pow, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, seed, n, out);
    return hipGetLastError();
}

template struct Launch<double>;

}  // namespace rtx

