// rt_device.h -- device-side arithmetic of the render loop, templated on the scalar type.
//
// One translation unit instantiates it for double with floating-point contraction OFF
// (the reference is Rust: every multiply and add rounds separately, no FMA), another for
// float with contraction on.  Operation order follows the reference expression by
// expression; each function names the Rust item it restates (paths relative to the
// reference checkout).  Two terms the reference evaluates are dropped because they can
// only change the sign of a zero, which nothing downstream observes: the `0.0 +` that
// starts every matrix row (src/matrices/matrix4.rs:179-181) and the `m[i][3] * 0.0` a
// vector contributes.  The fourth matrix row, computed and discarded by the reference
// (matrix4.rs:186-190), is never computed.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_types.h"

#define RT_DEV __device__ __forceinline__

namespace rtx {

// ---- scalar helpers -----------------------------------------------------------------
template <typename R> struct Num;
template <> struct Num<double> {
    static constexpr bool exact = true;                      // expression-by-expression restatement
    static constexpr double cull_slack = 1e-12;              // relative rounding allowance of the bounding-sphere test
    static constexpr double eps = 0.00001;                   // src/utils.rs:1
    static constexpr double big = 1.7976931348623157e308;    // f64::MAX
    static constexpr double inf = __builtin_huge_val();
    static RT_DEV double sqrt_(double x);  // below: the compiler's sequence without its range handling when x allows
    static RT_DEV double div(double a, double b) { return a / b; }
    static RT_DEV double pow_(double a, double b);  // below: pow_specular
    static RT_DEV double abs_(double x) { return __builtin_fabs(x); }
    static RT_DEV double floor_(double x) { return __builtin_floor(x); }
    static RT_DEV double max_(double a, double b) { return __builtin_fmax(a, b); }
    static RT_DEV double min_(double a, double b) { return __builtin_fmin(a, b); }
};
// The float instantiation is the fast path: it has no bit-parity contract (8-bit colours within
// 1/255 of the f64 result on >= 99.9 % of the pixels), so it uses the hardware's 1-ulp
// reciprocal, square root, exp2 and log2, and formulations that stay accurate in 24 bits.
template <> struct Num<float> {
    static constexpr bool exact = false;
    static constexpr float cull_slack = 1e-6f;  // md and tc * tc each carry about 3 roundings of 6e-8
    static constexpr float eps = 0.00001f;
    static constexpr float big = 3.402823466e38f;
    static constexpr float inf = __builtin_huge_valf();
    static RT_DEV float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
    static RT_DEV float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
    static RT_DEV float pow_(float a, float b) { return __builtin_amdgcn_exp2f(b * __builtin_amdgcn_logf(a)); }
    static RT_DEV float abs_(float x) { return __builtin_fabsf(x); }
    static RT_DEV float floor_(float x) { return __builtin_floorf(x); }
    static RT_DEV float max_(float a, float b) { return __builtin_fmaxf(a, b); }
    static RT_DEV float min_(float a, float b) { return __builtin_fminf(a, b); }
};

template <typename R> struct V3 { R x, y, z; };

// ---- several quotients by one divisor ----------------------------------------------------
// The compiler expands a / b in f64 into: v_div_scale of both operands (moves them away from the
// exponent extremes), r0 = rcp(b), four FMAs that refine r0 to r, then m = a * r, the residual
// e = fma(-b, m, a), the correction fma(e, r, m) (v_div_fmas) and v_div_fixup for NaN, infinity,
// zero and denormal operands.  When every operand is an ordinary number with an exponent in
// [-300, 300] the scaling is the identity and the fix-up passes the quotient through, so the
// steps up to r depend on b alone.  Computing them once and the last three per numerator gives
// the very bits of the separate divisions (correctly rounded quotients, as the reference's
// divisions are) for 5 + 3k instructions instead of 11k.  Operands outside that range take the
// plain divisions.
RT_DEV uint32_t exponent_bits(double v) { return (uint32_t)(__builtin_bit_cast(uint64_t, v) >> 32) & 0x7fffffffu; }
constexpr uint32_t kDivLo = (1023u - 300u) << 20, kDivHi = (1023u + 300u) << 20;
RT_DEV bool ordinary(uint32_t e) { return e - kDivLo < kDivHi - kDivLo; }
RT_DEV double refined_rcp(double b)
{
    const double r0 = __builtin_amdgcn_rcp(b);
    const double e0 = __builtin_fma(-b, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-b, r1, 1.0);
    return __builtin_fma(r1, e1, r1);
}
RT_DEV double quotient_with(double a, double b, double r)
{
    const double m = a * r;
    return __builtin_fma(__builtin_fma(-b, m, a), r, m);
}
RT_DEV uint32_t high_word(double v) { return (uint32_t)(__builtin_bit_cast(uint64_t, v) >> 32); }
RT_DEV void exponent_range(V3<double> a, uint32_t &lo, uint32_t &hi)
{
    const uint32_t ex = exponent_bits(a.x), ey = exponent_bits(a.y), ez = exponent_bits(a.z);
    lo = ex < ey ? (ex < ez ? ex : ez) : (ey < ez ? ey : ez);
    hi = ex > ey ? (ex > ez ? ex : ez) : (ey > ez ? ey : ez);
}
RT_DEV V3<double> div3(V3<double> a, double b)
{
    uint32_t lo, hi;
    exponent_range(a, lo, hi);
    if (lo >= kDivLo && hi < kDivHi && ordinary(exponent_bits(b))) {
        const double r = refined_rcp(b);
        return { quotient_with(a.x, b, r), quotient_with(a.y, b, r), quotient_with(a.z, b, r) };
    }
    return { a.x / b, a.y / b, a.z / b };
}
// The square root is expanded the same way: the operand is scaled by 2^256 when it is below
// 2^-767, y = rsq(x) is refined by the ten operations below, the result is scaled back, and zeros
// and +infinity pass through.  For a positive x with an exponent in [-600, 600] only the ten
// operations act; they give the bits of the full sequence (a correctly rounded root, as the
// reference's f64::sqrt is).
constexpr uint32_t kSqrtLo = (1023u - 600u) << 20, kSqrtHi = (1023u + 600u) << 20;
RT_DEV bool ordinary_radicand(double x) { return high_word(x) - kSqrtLo < kSqrtHi - kSqrtLo; }  // false for x < 0 too
RT_DEV double sqrt_steps(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}
RT_DEV double Num<double>::sqrt_(double x) { return ordinary_radicand(x) ? sqrt_steps(x) : __builtin_sqrt(x); }
// x^y of the specular term (factor = reflect_dot_eye.powf(shininess), src/scenes/lights.rs:70-72),
// for x > 0.  The reference's powf is the host C library's pow; no device pow reproduces its bits
// (the general-purpose one here is within 1 ulp of it and takes about 220 instructions, a tenth
// of a color_at node), and the contract for colours is 1e-9.  This one is 2^(y log2 x) in plain
// double arithmetic, about 60 instructions: log2 from the mantissa m in [sqrt(1/2), sqrt(2)) as
// 2 atanh((m - 1) / (m + 1)) / ln 2, eleven terms; 2^f for |f| <= 1/2 as a 13th-degree Taylor
// polynomial.  Its error relative to the true power is |y log2 x| * 5e-16 at most, and a result
// that is not negligible has |y log2 x| < 40, so the factor, which lies in [0, 1], is off by
// 1e-14 at most (measured against the general-purpose pow on the device: rtx_selfcheck_pow).
// NaN gives NaN, y = 0 gives 1, overflow +inf, underflow 0.
RT_DEV double pow_specular(double x, double y)
{
    double m = __builtin_amdgcn_frexp_mant(x);  // [1/2, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double s = (m - 1.0) * refined_rcp(m + 1.0);
    const double z = s * s;
    double p = 1.0 / 21.0;
    p = __builtin_fma(p, z, 1.0 / 19.0);
    p = __builtin_fma(p, z, 1.0 / 17.0);
    p = __builtin_fma(p, z, 1.0 / 15.0);
    p = __builtin_fma(p, z, 1.0 / 13.0);
    p = __builtin_fma(p, z, 1.0 / 11.0);
    p = __builtin_fma(p, z, 1.0 / 9.0);
    p = __builtin_fma(p, z, 1.0 / 7.0);
    p = __builtin_fma(p, z, 1.0 / 5.0);
    p = __builtin_fma(p, z, 1.0 / 3.0);
    p = __builtin_fma(p, z, 1.0);
    const double log2x = __builtin_fma(s * 2.8853900817779268, p, (double)e);  // 2 / ln 2
    const double t = y * log2x;
    const double tc = __builtin_fmin(__builtin_fmax(t, -1100.0), 1100.0);
    const double n = __builtin_rint(tc);
    const double g = (tc - n) * 0.69314718055994531;
    double q = 1.0 / 6227020800.0;
    q = __builtin_fma(q, g, 1.0 / 479001600.0);
    q = __builtin_fma(q, g, 1.0 / 39916800.0);
    q = __builtin_fma(q, g, 1.0 / 3628800.0);
    q = __builtin_fma(q, g, 1.0 / 362880.0);
    q = __builtin_fma(q, g, 1.0 / 40320.0);
    q = __builtin_fma(q, g, 1.0 / 5040.0);
    q = __builtin_fma(q, g, 1.0 / 720.0);
    q = __builtin_fma(q, g, 1.0 / 120.0);
    q = __builtin_fma(q, g, 1.0 / 24.0);
    q = __builtin_fma(q, g, 1.0 / 6.0);
    q = __builtin_fma(q, g, 0.5);
    q = __builtin_fma(q, g, 1.0);
    q = __builtin_fma(q, g, 1.0);
    double r = __builtin_ldexp(q, (int)n);
    r = t == t ? r : t;
    return y == 0.0 ? 1.0 : r;
}
RT_DEV double Num<double>::pow_(double a, double b) { return pow_specular(a, b); }
// a / |a| (src/matrices/tuples.rs:58-73) with one range check for the root and the quotients: a
// squared magnitude with an exponent in [-600, 600] has a root that is ordinary for the division
RT_DEV V3<double> normalized(V3<double> a, double m2, double *length = nullptr)
{
    uint32_t lo, hi;
    exponent_range(a, lo, hi);
    if (ordinary_radicand(m2) && lo >= kDivLo && hi < kDivHi) {
        const double mag = sqrt_steps(m2);
        if (length) *length = mag;
        const double r = refined_rcp(mag);
        return { quotient_with(a.x, mag, r), quotient_with(a.y, mag, r), quotient_with(a.z, mag, r) };
    }
    const double mag = __builtin_sqrt(m2);
    if (length) *length = mag;
    return { a.x / mag, a.y / mag, a.z / mag };
}
RT_DEV V3<float> normalized(V3<float> a, float m2, float *length = nullptr)
{
    const float r = __builtin_amdgcn_rsqf(m2);
    if (length) *length = m2 * r;
    return { a.x * r, a.y * r, a.z * r };
}
RT_DEV V3<float> div3(V3<float> a, float b)
{
    const float r = __builtin_amdgcn_rcpf(b);
    return { a.x * r, a.y * r, a.z * r };
}

// src/matrices/tuples.rs:145-154
template <typename R> RT_DEV R dot3(V3<R> a, V3<R> b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// src/matrices/tuples.rs:58-73: magnitude, then three divisions
template <typename R> RT_DEV V3<R> normalize3(V3<R> a)
{
    const R m2 = (a.x * a.x + a.y * a.y) + a.z * a.z;
    if constexpr (Num<R>::exact) {
        return normalized(a, m2);
    } else {
        const R r = __builtin_amdgcn_rsqf(m2);
        return { a.x * r, a.y * r, a.z * r };
    }
}
// src/matrices/matrix4.rs:172-193 for a point (w = 1) and a vector (w = 0); m = 3 rows x 4
// (M: a pointer to the matrix in any address space)
template <typename R, typename M> RT_DEV V3<R> xform_point(M m, V3<R> p)
{
    return { ((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3],
             ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
             ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11] };
}
template <typename R, typename M> RT_DEV V3<R> xform_vector(M m, V3<R> v)
{
    return { (m[0] * v.x + m[1] * v.y) + m[2] * v.z,
             (m[4] * v.x + m[5] * v.y) + m[6] * v.z,
             (m[8] * v.x + m[9] * v.y) + m[10] * v.z };
}

// The same two products for a matrix whose upper-left 3x3 is diagonal (SLOT_DIAGONAL).  The
// general expressions add m * y with m = +-0 to the diagonal term, which leaves it as it is
// (x + 0 == x; only a result that is itself zero can come out with the other sign, and no
// intersection routine divides by, or takes the sign of, such a coordinate), so these are the
// reference's values for every finite ray at 6 + 3 operations instead of 18 + 15.  A ray with an
// infinite or NaN component is the exception (0 * inf is NaN there and not here): the reference
// panics on such a ray (the sort of its intersections unwraps a partial_cmp, world.rs:105).
template <typename R, typename M> RT_DEV V3<R> diag_point(M m, V3<R> p)
{
    return { m[0] * p.x + m[3], m[5] * p.y + m[7], m[10] * p.z + m[11] };
}
template <typename R, typename M> RT_DEV V3<R> diag_vector(M m, V3<R> v) { return { m[0] * v.x, m[5] * v.y, m[10] * v.z }; }

// ---- ray/object intersection in object space -----------------------------------------
// Each routine calls emit(t) once per intersection, in the order the reference pushes them;
// emit(t, ok) stands for `if (ok) emit(t)` where the condition is cheap (the callers fold it into
// their selects instead of branching around a handful of instructions).
// Terms that depend on the ray origin alone (`c` of the quadratics, the slab numerators of
// the cube) come in as arguments: a secondary ray computes them per lane, while primary rays
// -- which all start at the camera -- read them from a table built once per frame.

// src/shapes/spheres.rs:24-42; c = (origin . origin) - 1
template <typename R, typename Emit> RT_DEV void isect_sphere(V3<R> o, V3<R> d, R c, Emit &&emit)
{
    if constexpr (!Num<R>::exact) {
        // b^2 - 4ac cancels catastrophically in 24 bits once the sphere is far from the ray
        // origin.  Go through the point of the ray closest to the centre instead: with
        // h = d.o and l = o - (h/a) d, the roots are -h/a -+ sqrt((1 - l.l) / a).
        const R a = dot3(d, d);
        const R inv_a = __builtin_amdgcn_rcpf(a);
        const R ha = dot3(d, o) * inv_a;
        const V3<R> l = { o.x - ha * d.x, o.y - ha * d.y, o.z - ha * d.z };
        const R rem = R(1) - dot3(l, l);
        if (!(rem < R(0))) {
            const R s = Num<R>::sqrt_(rem * inv_a);
            emit(-ha - s);
            emit(-ha + s);
        }
        return;
    }
    R a = dot3(d, d);
    R b = R(2) * dot3(d, o);
    R disc = b * b - ((R(4) * a) * c);
    if (!(disc < R(0))) {
        R sq = Num<R>::sqrt_(disc);
        R two_a = R(2) * a;
        emit(Num<R>::div(-b - sq, two_a));
        emit(Num<R>::div(-b + sq, two_a));
    }
}
// src/shapes/planes.rs:11-17; neg_oy = -origin.y
template <typename R, typename Emit> RT_DEV void isect_plane(R neg_oy, R dy, Emit &&emit)
{
    emit(Num<R>::div(neg_oy, dy), !(Num<R>::abs_(dy) < Num<R>::eps));  // a quotient by (nearly) zero is computed and dropped
}
// src/shapes/cubes.rs:31-51; lo_num = -1 - origin, hi_num = 1 - origin
template <typename R> RT_DEV void cube_axis(R lo_num, R hi_num, R direction, R &tmin, R &tmax)
{
    R lo, hi;
    if (!(Num<R>::abs_(direction) < Num<R>::eps)) {
        lo = Num<R>::div(lo_num, direction);
        hi = Num<R>::div(hi_num, direction);
    } else {
        lo = lo_num * Num<R>::big;
        hi = hi_num * Num<R>::big;
    }
    if (lo > hi) { R tmp = lo; lo = hi; hi = tmp; }
    tmin = lo;
    tmax = hi;
}
// src/shapes/cubes.rs:7-29
template <typename R, typename Emit> RT_DEV void isect_cube(V3<R> lo_num, V3<R> hi_num, V3<R> d, Emit &&emit)
{
    R xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
    cube_axis(lo_num.x, hi_num.x, d.x, xtmin, xtmax);
    cube_axis(lo_num.y, hi_num.y, d.y, ytmin, ytmax);
    if (ytmin > xtmax || xtmin > ytmax) return;
    cube_axis(lo_num.z, hi_num.z, d.z, ztmin, ztmax);
    R tmin = Num<R>::max_(Num<R>::max_(xtmin, ytmin), ztmin);
    R tmax = Num<R>::min_(Num<R>::min_(xtmax, ytmax), ztmax);
    if (tmin > tmax) return;
    emit(tmin);
    emit(tmax);
}
// src/shapes/cylinders.rs:66-86 and src/shapes/cones.rs:84-104.  The cone's cap test
// compares against |y| (cones.rs:103), the cylinder's against 1 (cylinders.rs:85).
template <typename R, bool CONE, typename Emit>
RT_DEV void isect_caps(V3<R> o, V3<R> d, R ymin, R ymax, bool closed, Emit &&emit)
{
    if (!closed || Num<R>::abs_(d.y) < Num<R>::eps) return;
    R t1 = Num<R>::div(ymin - o.y, d.y);
    {
        R x = o.x + (t1 * d.x);
        R z = o.z + (t1 * d.z);
        R lim = CONE ? Num<R>::abs_(ymin) : R(1);
        emit(t1, x * x + z * z <= lim);
    }
    R t2 = Num<R>::div(ymax - o.y, d.y);
    {
        R x = o.x + (t2 * d.x);
        R z = o.z + (t2 * d.z);
        R lim = CONE ? Num<R>::abs_(ymax) : R(1);
        emit(t2, x * x + z * z <= lim);
    }
}
// the wall roots shared by cylinder and cone: cylinders.rs:37-60, cones.rs:54-78
template <typename R, typename Emit>
RT_DEV bool isect_wall(R a, R b, R c, V3<R> o, V3<R> d, R ymin, R ymax, Emit &&emit)
{
    R disc = b * b - (R(4) * a) * c;
    if (disc < R(0)) return false;
    R sq = Num<R>::sqrt_(disc);
    R two_a = R(2) * a;
    R t0 = Num<R>::div(-b - sq, two_a);
    R t1 = Num<R>::div(-b + sq, two_a);
    if (t0 > t1) { R tmp = t0; t0 = t1; t1 = tmp; }
    R y0 = o.y + (t0 * d.y);
    emit(t0, ymin < y0 && y0 < ymax);
    R y1 = o.y + (t1 * d.y);
    emit(t1, ymin < y1 && y1 < ymax);
    return true;
}
// src/shapes/cylinders.rs:27-64; o2 = 2 * origin, c = origin.x^2 + origin.z^2 - 1
template <typename R, typename Emit>
RT_DEV void isect_cylinder(V3<R> o, V3<R> o2, R c, V3<R> d, R ymin, R ymax, bool closed, Emit &&emit)
{
    R a = d.x * d.x + d.z * d.z;
    if (a < Num<R>::eps) {
        isect_caps<R, false>(o, d, ymin, ymax, closed, emit);
        return;
    }
    if constexpr (!Num<R>::exact) {
        // the sphere's closest-point form in the xz plane
        const R inv_a = __builtin_amdgcn_rcpf(a);
        const R ha = (o.x * d.x + o.z * d.z) * inv_a;
        const R lx = o.x - ha * d.x, lz = o.z - ha * d.z;
        const R rem = R(1) - (lx * lx + lz * lz);
        if (rem < R(0)) return;
        const R s = Num<R>::sqrt_(rem * inv_a);
        const R t0 = -ha - s, t1 = -ha + s;
        const R y0 = o.y + t0 * d.y;
        emit(t0, ymin < y0 && y0 < ymax);
        const R y1 = o.y + t1 * d.y;
        emit(t1, ymin < y1 && y1 < ymax);
        isect_caps<R, false>(o, d, ymin, ymax, closed, emit);
        return;
    }
    R b = o2.x * d.x + o2.z * d.z;
    if (!isect_wall(a, b, c, o, d, ymin, ymax, emit)) return;
    isect_caps<R, false>(o, d, ymin, ymax, closed, emit);
}
// src/shapes/cones.rs:27-82; o2 = 2 * origin, c = origin.x^2 - origin.y^2 + origin.z^2
template <typename R, typename Emit>
RT_DEV void isect_cone(V3<R> o, V3<R> o2, R c, V3<R> d, R ymin, R ymax, bool closed, Emit &&emit)
{
    if constexpr (!Num<R>::exact) {
        // The cone's quadratic form is indefinite, so there is no closest-point rewrite; its
        // coefficients and roots are evaluated in f64 from the f32 object-space ray.
        const double ox = o.x, oy = o.y, oz = o.z, dx = d.x, dy = d.y, dz = d.z;
        const double a = (dx * dx - dy * dy) + dz * dz;
        const double b = 2.0 * ((ox * dx - oy * dy) + oz * dz);
        const double c = (ox * ox - oy * oy) + oz * oz;
        const bool a0 = __builtin_fabs(a) < 0.00001, b0 = __builtin_fabs(b) < 0.00001;
        if (!(a0 && b0)) {
            if (a0) {
                emit(R(-c / (2.0 * b)));
            } else {
                const double disc = b * b - 4.0 * a * c;
                if (disc < 0.0) return;
                const double sq = __builtin_sqrt(disc);
                double t0 = (-b - sq) / (2.0 * a), t1 = (-b + sq) / (2.0 * a);
                if (t0 > t1) { const double tmp = t0; t0 = t1; t1 = tmp; }
                const double y0 = oy + t0 * dy;
                emit(R(t0), ymin < y0 && y0 < ymax);
                const double y1 = oy + t1 * dy;
                emit(R(t1), ymin < y1 && y1 < ymax);
            }
        }
        isect_caps<R, true>(o, d, ymin, ymax, closed, emit);
        return;
    }
    R a = (d.x * d.x - d.y * d.y) + d.z * d.z;
    R b = (o2.x * d.x - o2.y * d.y) + o2.z * d.z;
    bool a0 = Num<R>::abs_(a) < Num<R>::eps;
    bool b0 = Num<R>::abs_(b) < Num<R>::eps;
    if (a0 && b0) {
        isect_caps<R, true>(o, d, ymin, ymax, closed, emit);
        return;
    }
    if (a0) {
        emit(Num<R>::div(-c, R(2) * b));
        isect_caps<R, true>(o, d, ymin, ymax, closed, emit);
        return;
    }
    if (!isect_wall(a, b, c, o, d, ymin, ymax, emit)) return;
    isect_caps<R, true>(o, d, ymin, ymax, closed, emit);
}
// the origin-only terms, as the reference's expressions evaluate them
template <typename R> RT_DEV R sphere_c(V3<R> o) { return dot3(o, o) - R(1); }
template <typename R> RT_DEV R cylinder_c(V3<R> o) { return (o.x * o.x + o.z * o.z) - R(1); }
template <typename R> RT_DEV R cone_c(V3<R> o) { return (o.x * o.x - o.y * o.y) + o.z * o.z; }
template <typename R> RT_DEV V3<R> twice(V3<R> o) { return { R(2) * o.x, R(2) * o.y, R(2) * o.z }; }

// ---- bounding-sphere tests ---------------------------------------------------------------
// In scenes with many objects nearly every ray misses nearly every object, and the exact
// per-shape test (50-150 instructions, more with its divisions) is spent on proving that.  A
// test against the object's world-space bounding sphere costs about 20 and needs no transform;
// when no lane of the wave can hit the object the record is not even fetched.  The spheres are
// inflated on the host and the comparisons carry a rounding allowance, so a skipped object is
// one for which the exact test would have found no intersection (in the range that counts):
// results do not change.
// The test's own arithmetic is not the reference's: it uses fused multiply-adds and a unit
// direction, whatever the translation unit's contraction setting, to keep it short.
RT_DEV double cull_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
RT_DEV float cull_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <typename R>
struct CullRay {
    V3<R> o, u;  // a point of the ray's line, unit direction
    R dlen;      // |d|: t of the ray times dlen is distance along u
    R shift;     // o lies this far along u from the ray's origin (0: o is the origin)
    bool dead;   // no object with a bounding sphere can be hit (a NaN in the ray, or all of them behind it)
};
// Origins far from the scene make the sphere tests useless: the squared distance of a centre from
// the line is a difference of two numbers of size |centre - origin|^2, and the allowance for its
// rounding (and for that of the exact test, which grows the same way) then exceeds every radius.
// A ray reflected towards the horizon meets a wall plane two million units away; its shadow ray
// starts there, and in 24 bits every sphere of the scene "may be hit": such a wave walked all
// 10 000 objects for a millisecond.  The fast path therefore tests a ray that starts outside the
// sphere around all bounded objects from the point where its line reaches that sphere: the same
// line up to the rounding of the move, which is no worse than that of the fast path's own
// intersection tests at such an origin.  The exact path keeps the origin: there the allowance has
// to cover the reference's arithmetic, which is evaluated from the true origin.
template <typename R, bool LINE> RT_DEV CullRay<R> cull_ray(const SceneDev<R> &S, V3<R> o, V3<R> d)
{
    CullRay<R> c;
    c.o = o;
    c.shift = R(0);
    const R d2 = dot3(d, d);
    c.dlen = Num<R>::sqrt_(d2);
    // a zero or infinite direction gives NaNs here, with which no comparison below skips
    const R inv = R(1) / c.dlen;
    c.u = { d.x * inv, d.y * inv, d.z * inv };
    // A NaN anywhere in the ray reaches every object-space coordinate (0 * NaN is NaN), so every
    // t of every shape is NaN and nothing is hit; without this the sphere tests, whose
    // comparisons are all false on NaN, would let such a ray through to all objects.
    const R oo = dot3(o, o);
    c.dead = !(d2 == d2) || !(oo == oo);
    const V3<R> m = { S.world_c[0] - o.x, S.world_c[1] - o.y, S.world_c[2] - o.z };
    const R m2 = cull_fma(m.z, m.z, cull_fma(m.y, m.y, m.x * m.x));
    if (m2 > S.world_r * S.world_r) {  // the origin is outside (never true of a NaN or an infinite radius)
        const R tc = cull_fma(m.z, c.u.z, cull_fma(m.y, c.u.y, m.x * c.u.x));
        if constexpr (!LINE) c.dead = c.dead || tc < R(0);  // t^2 - 2 t tc + (m2 - r^2) > 0 for every t >= 0: the ray never enters
        if constexpr (!Num<R>::exact) {
            // Every point of the sphere is at least tc - world_r along u from the origin (before it, for
            // intersections anywhere on the line): nothing bounded lies between the two points.
            const R s = tc - S.world_r;
            if (Num<R>::abs_(s) < Num<R>::big) {  // false for NaN
                c.o = { cull_fma(c.u.x, s, o.x), cull_fma(c.u.y, s, o.y), cull_fma(c.u.z, s, o.z) };
                c.shift = s;
            }
        }
    }
    return c;
}
// LINE: intersections anywhere on the line count (the container bookkeeping of refraction reads
// those behind the origin too).  Otherwise only 0 <= t <= t_max counts; reach = t_max * |d|.
// Three quantities, any of which being positive rules the sphere out, joined by a maximum and
// compared once (no branches, one mask: the walks spend as many scalar instructions as vector
// ones).  A NaN on one side of a maximum or minimum gives the other side and NaN > 0 is false; the
// only case in which that differs from three separate comparisons is an origin at infinity,
// whose exact intersections are all NaN as well.
template <typename R, bool LINE>
RT_DEV bool may_hit(const BoundRec<R> &b, const CullRay<R> &cr, R reach)
{
    const V3<R> m = { b.c[0] - cr.o.x, b.c[1] - cr.o.y, b.c[2] - cr.o.z };
    const R tc = cull_fma(m.z, cr.u.z, cull_fma(m.y, cr.u.y, m.x * cr.u.x));
    const R m2 = cull_fma(m.z, m.z, cull_fma(m.y, m.y, m.x * m.x));
    const R r2 = b.r * b.r;
    // squared distance of the centre from the line against r^2; r = inf never skips
    R out = cull_fma(-Num<R>::cull_slack, m2, cull_fma(-tc, tc, m2)) - r2;
    if constexpr (!LINE) {
        const R behind = Num<R>::min_(-tc, m2 - r2);  // both positive: the whole sphere lies behind the origin
        // the ray is inside the sphere no earlier than tc - r along u
        const R far = cull_fma(-Num<R>::cull_slack, Num<R>::abs_(tc), tc - b.r) - reach;
        out = Num<R>::max_(Num<R>::max_(out, behind), far);
    }
    return !cr.dead & !(out > R(0));
}

// ---- walking the object tables ---------------------------------------------------------
// Visit records 0..n-1 of one table with a wave-uniform index.  The table is read through a
// constant-address-space pointer: that is what makes the compiler fetch each record once per
// wave through the scalar cache into SGPRs even inside loops that also store to memory (with a
// generic pointer it falls back to one vector load per lane there).  The tables are written
// before the launch and never by a kernel that walks them.  The other waves of the SIMD cover
// the fetch latency.  keep(slot) says whether this lane's ray can hit the object at all; the
// record is fetched only when some lane's can.

#define RT_CONSTANT_AS __attribute__((address_space(4)))
// One value of a table that no kernel writes while it is read (object tables, and the queue
// lengths the previous launch left): through the constant address space, so that a wave-uniform
// address becomes a scalar load even after stores to other memory.
template <typename T> RT_DEV T ro_load(const T *p) { return *(const RT_CONSTANT_AS T *)p; }
template <typename Rec> RT_DEV Rec load_uniform(const Rec *p)
{
    Rec rec;
    __builtin_memcpy(&rec, (const RT_CONSTANT_AS Rec *)p, sizeof(Rec));
    return rec;
}
// keep.object(slot) says whether this lane's ray can hit the object at all; the record is fetched
// only when some lane's can.  With Keep::clustered the slots are visited group by group and
// cluster by cluster, and keep.group(g) / keep.cluster(c) are asked first.
// group_order: the order in which to meet the table's groups (table-relative indices), or null for table order
template <typename Rec, typename Keep, typename Body>
RT_DEV void walk_table(const Rec *recs, uint32_t n, uint32_t slot0, uint32_t cluster0, uint32_t group0, const Keep &keep,
                       Body &&body, const uint32_t *group_order = nullptr)
{
    if constexpr (Keep::clustered) {
        // Spheres are fetched four at a time: a walk is a chain of scalar loads, each of which
        // costs its full latency when the wave is alone on its SIMD (secondary rays), and one
        // wait then covers four of them.  kCluster is a multiple of 4; the tables are padded.
        static_assert(kCluster % 4 == 0, "spheres are fetched in fours");
        const uint32_t n_cl = (n + kCluster - 1) / kCluster;
        const uint32_t n_grp = (n_cl + kCluster - 1) / kCluster;
        for (uint32_t turn = 0; turn < n_grp; ++turn) {
            const uint32_t g_rel = group_order ? ro_load(group_order + turn) : turn;
            const uint32_t g = group0 + g_rel, c0 = g_rel * kCluster;
            const bool kg = keep.group(g);
            if (__ballot(kg) == 0ull) continue;
            const uint32_t c1 = (c0 + kCluster < n_cl) ? c0 + kCluster : n_cl;
            for (uint32_t cb = c0; cb < c1; cb += 4) {
                bool kc4[4];
                keep.cluster4(cluster0 + cb, kg, kc4);
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t c = cb + j;
                    if (c >= c1 || __ballot(kc4[j]) == 0ull) continue;
                    const uint32_t i0 = c * kCluster;
                    const uint32_t i1 = (i0 + kCluster < n) ? i0 + kCluster : n;
                    for (uint32_t ib = i0; ib < i1; ib += 4) {
                        bool ko4[4];
                        keep.object4(slot0 + ib, kc4[j], ko4);
                        for (uint32_t e = 0; e < 4; ++e) {
                            const uint32_t i = ib + e;
                            if (i >= i1 || __ballot(ko4[e]) == 0ull) continue;
                            const Rec rec = load_uniform(recs + i);
                            if (ko4[e]) body(rec, slot0 + i);
                        }
                    }
                }
            }
        }
    } else {
        for (uint32_t i = 0; i < n; ++i) {
            const Rec rec = load_uniform(recs + i);
            body(rec, slot0 + i);
        }
    }
}
struct KeepAll {
    static constexpr bool clustered = false;
};
// Bounding-sphere culling for one ray.  `best`: the closest hit so far (intersections beyond it
// cannot matter), `limit`: a fixed upper bound on top of that, `stop`: set once the walk's
// answer is known (any-hit shadow rays).
template <typename R, bool LINE>
struct SphereCull {
    static constexpr bool clustered = true;
    const BoundRec<R> *bnd, *cl, *grp;
    CullRay<R> cr;
    const R *best;
    R limit;
    const bool *stop;
    RT_DEV R reach() const { return (best ? Num<R>::min_(*best, limit) : limit) * cr.dlen - cr.shift; }  // from cr.o
    struct Four {
        BoundRec<R> b[4];
    };
    // four consecutive spheres in one fetch; entries past the end of the table are padding and
    // their answers are ignored by the caller
    RT_DEV void four(const BoundRec<R> *first, bool outer, bool *out) const
    {
        const Four f = load_uniform(reinterpret_cast<const Four *>(first));
        const bool live = outer && !(stop && *stop);
        const R rch = reach();
        for (int j = 0; j < 4; ++j) out[j] = live & may_hit<R, LINE>(f.b[j], cr, rch);
    }
    RT_DEV void object4(uint32_t slot, bool outer, bool *out) const { four(bnd + slot, outer, out); }
    RT_DEV void cluster4(uint32_t c, bool outer, bool *out) const { four(cl + c, outer, out); }
    RT_DEV bool group(uint32_t g) const { return !(stop && *stop) && may_hit<R, LINE>(load_uniform(grp + g), cr, reach()); }
    // one sphere for this lane alone (the index differs from lane to lane)
    RT_DEV bool lane_group(uint32_t g) const { return may_hit<R, LINE>(grp[g], cr, reach()); }
    RT_DEV bool lane_cluster(uint32_t c) const { return may_hit<R, LINE>(cl[c], cr, reach()); }
    RT_DEV bool lane_object(uint32_t slot) const { return may_hit<R, LINE>(bnd[slot], cr, reach()); }
};

// The same tests for rays that all start at the camera.  Of the 19 operations of may_hit, 10 work
// out things that depend on the sphere and the origin alone (centre - origin, its square, r^2);
// make_primary_bound does those once per frame and sphere, and a primary ray is left with the dot
// product tc = m.u and three quantities of which any being positive rules the sphere out:
//   k - tc^2            the line misses it; k = (1 - allowance) m2 - r^2, which is the test of
//                       may_hit up to two more roundings of size ulp(m2), inside the allowance
//   behind - tc         the camera is outside it and it lies behind
//   tc - r - allowance |tc| - reach   it begins beyond the closest hit so far
// They are joined by a maximum and compared once: a camera wave spends as many scalar
// instructions as vector ones, and a test written as three short-circuit comparisons costs a
// dozen of them in mask bookkeeping and branches.  A NaN on one side of a maximum gives the other
// side and NaN > 0 is false, as with the separate comparisons; r = inf makes all three -inf.
// The origin is not moved towards the scene here (CullRay::shift is ignored): all of it is
// relative to the camera itself.
template <typename R>
RT_DEV void make_primary_bound(const SceneDev<R> &S, const ApexBounds<R> &to, V3<R> cam_origin, uint32_t i)
{
    const BoundRec<R> *src;
    PrimBoundRec<R> *dst;
    if (i < S.n_bnd_total) { src = S.bnd + i; dst = const_cast<PrimBoundRec<R> *>(to.bnd) + i; }
    else if ((i -= S.n_bnd_total) < S.n_cl_total) { src = S.cl + i; dst = const_cast<PrimBoundRec<R> *>(to.cl) + i; }
    else if ((i -= S.n_cl_total) < S.n_grp_total) { src = S.grp + i; dst = const_cast<PrimBoundRec<R> *>(to.grp) + i; }
    else return;
    const BoundRec<R> b = *src;
    PrimBoundRec<R> out;
    out.m[0] = b.c[0] - cam_origin.x;
    out.m[1] = b.c[1] - cam_origin.y;
    out.m[2] = b.c[2] - cam_origin.z;
    const R m2 = cull_fma(out.m[2], out.m[2], cull_fma(out.m[1], out.m[1], out.m[0] * out.m[0]));
    const R r2 = b.r * b.r;
    out.k = cull_fma(-Num<R>::cull_slack, m2, m2) - r2;
    out.r = b.r;
    out.behind = m2 > r2 ? R(0) : -Num<R>::inf;
    out.km = cull_fma(R(-4) * Num<R>::cull_slack, m2, m2);
    out.dist = Num<R>::sqrt_(out.km) * (R(1) - Num<R>::cull_slack);
    *dst = out;
}
// what one ray's test reads of a PrimBoundRec
template <typename R>
struct alignas(sizeof(R) * 2) PrimBoundHead {
    R m[3], k, r, behind;
};
template <typename R, bool LINE>
RT_DEV bool may_hit_primary(const PrimBoundHead<R> &b, V3<R> u, R reach)
{
    const R tc = cull_fma(b.m[2], u.z, cull_fma(b.m[1], u.y, b.m[0] * u.x));
    R out = cull_fma(-tc, tc, b.k);
    if constexpr (!LINE) {
        const R far = cull_fma(-Num<R>::cull_slack, Num<R>::abs_(tc), tc - b.r) - reach;
        out = Num<R>::max_(Num<R>::max_(out, b.behind - tc), far);
    }
    return !(out > R(0));
}
template <typename R, bool LINE>
struct PrimSphereCull {
    static constexpr bool clustered = true;
    const PrimBoundRec<R> *bnd, *cl, *grp;
    V3<R> u;
    R dlen;
    bool dead;  // asked once per group: clusters and objects are reached through a group that passed
    const R *best;
    RT_DEV R reach() const { return *best * dlen; }
    static RT_DEV PrimBoundHead<R> head(const PrimBoundRec<R> *rec) { return load_uniform(reinterpret_cast<const PrimBoundHead<R> *>(rec)); }
    RT_DEV void four(const PrimBoundRec<R> *first, bool outer, bool *out) const
    {
        PrimBoundHead<R> b[4];
        for (int j = 0; j < 4; ++j) b[j] = head(first + j);
        const R rch = reach();
        for (int j = 0; j < 4; ++j) out[j] = outer & may_hit_primary<R, LINE>(b[j], u, rch);
    }
    RT_DEV void object4(uint32_t slot, bool outer, bool *out) const { four(bnd + slot, outer, out); }
    RT_DEV void cluster4(uint32_t c, bool outer, bool *out) const { four(cl + c, outer, out); }
    RT_DEV bool group(uint32_t g) const { return !dead & may_hit_primary<R, LINE>(head(grp + g), u, reach()); }
};

// The walk for a wave that holds ONE ray (the deep recursion levels of a scene with thousands of
// objects hold a few thousand rays for a chip of 65 000 lanes): the ray is the same in every lane
// and the 64 lanes test 64 spheres of a level at once -- every group of a table, then the 16
// clusters of up to four groups that may be hit, then the 16 members of up to four such
// clusters, each lane finishing with the exact intersection of the one object it is left with.
// Some 30 steps where the wave-uniform walk with one lane in use takes 500 dependent fetches.
// Every lane keeps the closest hit among the objects it tested itself (closest_hit_with joins
// them); what a finished batch found shortens the reach of the next (wave_best).
// Same tests, same spheres, same exact intersections as the other walks; the order in which
// objects are met differs, and ties between equal t are settled by World index, not by order.
// WALK_WAVE, WALK_WIDE: bounding spheres fetched from memory, a wave at a time for its 64 rays, or by a whole wave
// for one ray; WALK_LANE, WALK_TREE_WIDE: the packed hierarchy in LDS, every lane for its own ray, or a whole wave
// for one ray.  (A wave at a time over the packed hierarchy was measured too: a wave of 64 unrelated rays visits
// most of a scene's clusters whatever the spheres are fetched from -- 2.3 against 0.83 ms on the first level of a
// 4K frame of 10 002 objects.)
enum : int { WALK_WAVE = 0, WALK_LANE = 1, WALK_WIDE = 2, WALK_TREE_WIDE = 4 };
constexpr bool walk_in_lds(int walk) { return walk == WALK_LANE || walk == WALK_TREE_WIDE; }
constexpr bool walk_is_wide(int walk) { return walk == WALK_WIDE || walk == WALK_TREE_WIDE; }
template <typename T> RT_DEV T wave_min(T v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const T other = __shfl_xor(v, off);
        v = other < v ? other : v;
    }
    return v;
}
// the positions of the (up to) four lowest set bits of a wave-uniform mask, which loses them
RT_DEV uint32_t pop_four(unsigned long long &mask, uint32_t pos[4])
{
    uint32_t count = 0;
    for (int j = 0; j < 4; ++j) {
        pos[j] = 0u;
        if (mask != 0ull) {
            pos[j] = (uint32_t)__builtin_ctzll(mask);
            mask &= mask - 1ull;
            ++count;
        }
    }
    return count;
}
template <typename R, typename Keep, typename Rec, typename Body>
RT_DEV void wide_walk_table(const Rec *recs, uint32_t n, uint32_t slot0, uint32_t cluster0, uint32_t group0,
                            const Keep &keep, const R *lane_best, R *wave_best, Body &&body)
{
    static_assert(kCluster == 16, "a quarter of the wave per cluster");
    if (n == 0u) return;
    if (keep.stop && __ballot(*keep.stop) != 0ull) return;
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t quarter = lane >> 4, member = lane & 15u;
    const uint32_t n_cl = (n + kCluster - 1) / kCluster, n_grp = (n_cl + kCluster - 1) / kCluster;
    auto mine = [&](const uint32_t v[4]) { return quarter == 0u ? v[0] : quarter == 1u ? v[1] : quarter == 2u ? v[2] : v[3]; };
    for (uint32_t g0 = 0; g0 < n_grp; g0 += 64u) {
        bool hit_g = false;
        if (g0 + lane < n_grp) hit_g = keep.lane_group(group0 + g0 + lane);
        unsigned long long groups = __ballot(hit_g);
        while (groups != 0ull) {
            uint32_t gpos[4];
            const uint32_t n_g = pop_four(groups, gpos);
            const uint32_t c = (g0 + mine(gpos)) * kCluster + member;
            bool hit_c = false;
            if (quarter < n_g && c < n_cl) hit_c = keep.lane_cluster(cluster0 + c);
            unsigned long long clusters = __ballot(hit_c);
            while (clusters != 0ull) {
                uint32_t cpos[4], cidx[4];
                const uint32_t n_c = pop_four(clusters, cpos);
                for (int j = 0; j < 4; ++j) cidx[j] = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)cpos[j]);  // the cluster lane cpos[j] tested
                const uint32_t i = mine(cidx) * kCluster + member;
                bool hit_o = false;
                if (quarter < n_c && i < n) hit_o = keep.lane_object(slot0 + i);
                if (hit_o) {
                    const Rec rec = recs[i];
                    body(rec, slot0 + i);
                }
                if (__ballot(hit_o) != 0ull) {
                    if (keep.stop && __ballot(*keep.stop) != 0ull) return;
                    if (lane_best) *wave_best = wave_min(*lane_best);
                }
            }
        }
    }
}

// The walk for secondary rays that have nothing to do with each other (reflected and refracted rays
// of a scene of thousands of objects, and the shadow rays of their hits): every lane walks the
// hierarchy for its own ray.  The wave-uniform walk visits the union of what 64 such rays need --
// most of the scene's groups, hundreds of clusters -- one dependent scalar fetch (a microsecond
// under load) after another.  Here the hierarchy is the packed copy a workgroup keeps in LDS
// (QuantNode: 8 bytes a sphere, stage_lane_tree), a lane reads the sphere its own position names at
// LDS latency, and only what the packed spheres let through is fetched from memory and intersected
// exactly -- per lane, a different object in every lane, with the expressions of the other walks.
// The packed spheres hold the stored ones (rounded outwards on the host), the tests are may_hit's
// in single precision on a ray moved to where it enters the sphere around all bounded objects, and a
// ray's own allowance covers what the exact test's rounding grows to when the true origin is far
// away (as the allowance of the exact path does): the objects that reach an exact intersection
// include all that can be hit, so hits, colours and counters are those of the other walks.
#define RT_LDS_AS __attribute__((address_space(3)))
template <typename R>
struct LaneTreeCull {
    static constexpr bool clustered = true;
    const RT_LDS_AS unsigned long long *nodes;  // the packed hierarchy in LDS: objects by slot, then clusters, then groups
    uint32_t cl0, grp0;            // node index of the first cluster and of the first group sphere
    float base[3], step;
    CullRay<float> cr;
    float extra_r;                 // added to every radius: sqrt of the exact test's allowance for this ray
    R dlen, shift;                 // |direction| and how far along it cr.o lies from the ray's origin
    const R *best;
    R limit;
    R limit_closed;                // the same for spheres and cubes (containers_behind; elsewhere = limit)
    const bool *stop;
    RT_DEV float reach() const
    {
        const R t = best ? Num<R>::min_(*best, limit) : limit;
        const float r = (float)(t * dlen - shift);
        return __builtin_fmaf(__builtin_fabsf(r), 4e-7f, r) + extra_r;  // rounded up; inf stays inf
    }
    RT_DEV bool test(uint32_t node, float rch) const
    {
        const unsigned long long q = nodes[node];
        const uint32_t lo = (uint32_t)q, hi = (uint32_t)(q >> 32);
        BoundRec<float> b;
        b.c[0] = __builtin_fmaf((float)(lo & 0xffffu), step, base[0]);
        b.c[1] = __builtin_fmaf((float)(lo >> 16), step, base[1]);
        b.c[2] = __builtin_fmaf((float)(hi & 0xffffu), step, base[2]);
        const uint32_t qr = hi >> 16;
        b.r = qr == 0xffffu ? Num<float>::inf : __builtin_fmaf((float)qr, step, extra_r);
        return may_hit<float, false>(b, cr, rch);
    }
    // Two neighbouring spheres at once, two floats to an instruction (v_pk_*_f32: the one place where the
    // single-precision vector unit does two operations per lane and cycle): bit 0 of the answer for `node`,
    // bit 1 for `node + 1`.  The expressions are may_hit's for rays that count from their origin on.
    typedef float Pair __attribute__((ext_vector_type(2)));
    RT_DEV uint32_t test_pair(uint32_t node, float rch) const
    {
        const unsigned long long qa = nodes[node], qb = nodes[node + 1u];
        const uint32_t la = (uint32_t)qa, ha = (uint32_t)(qa >> 32), lb = (uint32_t)qb, hb = (uint32_t)(qb >> 32);
        const Pair qx = { (float)(la & 0xffffu), (float)(lb & 0xffffu) }, qy = { (float)(la >> 16), (float)(lb >> 16) },
                   qz = { (float)(ha & 0xffffu), (float)(hb & 0xffffu) }, qr = { (float)(ha >> 16), (float)(hb >> 16) };
        const Pair st = { step, step };
        // centre - origin of the ray, and the radius
        const Pair mx = __builtin_elementwise_fma(qx, st, Pair{ base[0] - cr.o.x, base[0] - cr.o.x });
        const Pair my = __builtin_elementwise_fma(qy, st, Pair{ base[1] - cr.o.y, base[1] - cr.o.y });
        const Pair mz = __builtin_elementwise_fma(qz, st, Pair{ base[2] - cr.o.z, base[2] - cr.o.z });
        Pair r = __builtin_elementwise_fma(qr, st, Pair{ extra_r, extra_r });
        r.x = (ha >> 16) == 0xffffu ? Num<float>::inf : r.x;
        r.y = (hb >> 16) == 0xffffu ? Num<float>::inf : r.y;
        const Pair ux = { cr.u.x, cr.u.x }, uy = { cr.u.y, cr.u.y }, uz = { cr.u.z, cr.u.z };
        const Pair tc = __builtin_elementwise_fma(mz, uz, __builtin_elementwise_fma(my, uy, mx * ux));
        const Pair m2 = __builtin_elementwise_fma(mz, mz, __builtin_elementwise_fma(my, my, mx * mx));
        const Pair r2 = r * r;
        const Pair slack = { -Num<float>::cull_slack, -Num<float>::cull_slack };
        const Pair out = __builtin_elementwise_fma(slack, m2, __builtin_elementwise_fma(-tc, tc, m2)) - r2;
        const Pair behind = __builtin_elementwise_min(-tc, m2 - r2);
        const Pair far_ = __builtin_elementwise_fma(slack, __builtin_elementwise_abs(tc), tc - r) - Pair{ rch, rch };
        const Pair worst = __builtin_elementwise_max(__builtin_elementwise_max(out, behind), far_);
        return cr.dead ? 0u : ((worst.x > 0.0f ? 0u : 1u) | (worst.y > 0.0f ? 0u : 2u));
    }
    RT_DEV bool stopped() const { return stop && *stop; }
    // for the walk of a whole wave (walk_table): the same sphere for every lane, its own answer for each
    RT_DEV bool group(uint32_t g) const { return !stopped() && test(grp0 + g, reach()); }
    RT_DEV void four(uint32_t first, bool outer, bool *out) const
    {
        const bool live = outer && !stopped();
        const float rch = reach();
        for (uint32_t j = 0; j < 4u; ++j) out[j] = live & test(first + j, rch);
    }
    RT_DEV void cluster4(uint32_t c, bool outer, bool *out) const { four(cl0 + c, outer, out); }
    RT_DEV void object4(uint32_t slot, bool outer, bool *out) const { four(slot, outer, out); }
    // for the walk of one ray by a whole wave (wide_walk_table)
    RT_DEV bool lane_group(uint32_t g) const { return test(grp0 + g, reach()); }
    RT_DEV bool lane_cluster(uint32_t c) const { return test(cl0 + c, reach()); }
    RT_DEV bool lane_object(uint32_t slot) const { return test(slot, reach()); }
};
template <typename R>
RT_DEV LaneTreeCull<R> lane_tree_cull(const SceneDev<R> &S, V3<R> o, V3<R> d, const R *best, R limit, const bool *stop)
{
    LaneTreeCull<R> k;
    k.nodes = (const RT_LDS_AS unsigned long long *)(uintptr_t)S.q_lds;
    k.cl0 = S.n_bnd_total;
    k.grp0 = S.n_bnd_total + S.n_cl_total;
    k.base[0] = S.q_base[0]; k.base[1] = S.q_base[1]; k.base[2] = S.q_base[2];
    k.step = S.q_step;
    k.best = best;
    k.limit = limit;
    k.limit_closed = limit;
    k.stop = stop;
    // the ray in the scalar type of the kernel: unit direction, and the point where its line enters the
    // sphere around all bounded objects when it starts outside (nothing bounded lies before that point)
    const R d2 = dot3(d, d);
    k.dlen = Num<R>::sqrt_(d2);
    const R inv = R(1) / k.dlen;
    const V3<R> u = { d.x * inv, d.y * inv, d.z * inv };
    const R oo = dot3(o, o);
    k.cr.dead = !(d2 == d2) || !(oo == oo);  // a NaN in the ray: every t of every shape is NaN, nothing is hit
    const V3<R> m = { S.world_c[0] - o.x, S.world_c[1] - o.y, S.world_c[2] - o.z };
    const R m2 = dot3(m, m);
    V3<R> from = o;
    k.shift = R(0);
    if (m2 > S.world_r * S.world_r) {  // never true of a NaN or of an infinite radius
        const R tc = dot3(m, u);
        k.cr.dead = k.cr.dead || tc < R(0);  // the whole sphere lies behind the origin
        const R s = tc - S.world_r;
        if (Num<R>::abs_(s) < Num<R>::big) {
            from = { o.x + u.x * s, o.y + u.y * s, o.z + u.z * s };
            k.shift = s;
        }
    }
    k.cr.o = { (float)from.x, (float)from.y, (float)from.z };
    k.cr.u = { (float)u.x, (float)u.y, (float)u.z };
    k.cr.dlen = 1.0f;
    k.cr.shift = 0.0f;
    // The exact intersection is evaluated from the true origin, and its rounding grows with the squared
    // distance of an object from there (the allowance of may_hit in the exact path: cull_slack * m2):
    // the same allowance here, as an addition to the radius, with the farthest bounded object's distance.
    k.extra_r = 0.0f;
    if constexpr (Num<R>::exact) {
        const R far_ = Num<R>::sqrt_(m2) + S.world_r;
        const float e = (float)(far_ * R(1.0e-6));  // sqrt(cull_slack) * distance
        k.extra_r = e == e ? e * 1.0001f : Num<float>::inf;
    }
    return k;
}
// The cull state a walk uses for the table of one shape.
template <typename Keep> RT_DEV const Keep &keep_for_shape(const Keep &k, uint32_t) { return k; }
template <typename R> RT_DEV LaneTreeCull<R> keep_for_shape(const LaneTreeCull<R> &k, uint32_t shape)
{
    LaneTreeCull<R> out = k;
    if (shape == SH_SPHERE || shape == SH_CUBE) out.limit = k.limit_closed;
    return out;
}
// Every thread of the workgroup: the packed hierarchy from memory into LDS (ends with a barrier).
template <typename R> RT_DEV void stage_lane_tree(const SceneDev<R> &S, RT_LDS_AS unsigned long long *lds, uint32_t n_threads)
{
    // two nodes per lane and step (the table is padded to an even count and 16-byte aligned)
    const uint4 *from = reinterpret_cast<const uint4 *>(S.q_nodes);
    RT_LDS_AS uint4 *to = (RT_LDS_AS uint4 *)lds;
    const uint32_t n2 = (S.q_n + 1u) / 2u;
#pragma unroll 4
    for (uint32_t i = threadIdx.x; i < n2; i += n_threads) {
        const uint4 v = from[i];
        to[i].x = v.x; to[i].y = v.y; to[i].z = v.z; to[i].w = v.w;
    }
    __syncthreads();
}
// One shape table for one ray per lane.  keep.test() is asked for the groups of the table (all lanes
// together: the same group for every lane, its own answer for each), then per lane for the 16 clusters
// of each group its ray may hit, the 16 members of each such cluster, and body(rec, slot) runs for the
// members that are left, one at a time, with the record fetched by the lane itself.
template <typename R, typename Rec, typename Body>
RT_DEV void lane_tree_walk_table(const Rec *recs, uint32_t n, uint32_t slot0, uint32_t cluster0, uint32_t group0,
                                 const LaneTreeCull<R> &keep, Body &&body)
{
    static_assert(kCluster == 16, "sixteen answers per mask");
    if (n == 0u || keep.cr.dead) return;
    const uint32_t n_cl = (n + kCluster - 1) / kCluster, n_grp = (n_cl + kCluster - 1) / kCluster;
    for (uint32_t g0 = 0; g0 < n_grp; g0 += 64u) {
        if (keep.stopped()) return;
        unsigned long long groups = 0ull;
        {
            const float rch = keep.reach();
            const uint32_t g1 = g0 + 64u < n_grp ? g0 + 64u : n_grp;
            for (uint32_t g = g0; g < g1; ++g)
                if (keep.test(keep.grp0 + group0 + g, rch)) groups |= 1ull << (g - g0);
        }
        while (groups != 0ull) {
            const uint32_t g = g0 + (uint32_t)__builtin_ctzll(groups);
            groups &= groups - 1ull;
            uint32_t clusters = 0u;
            {
                const float rch = keep.reach();
                const uint32_t c0 = g * kCluster;
#pragma unroll
                for (uint32_t j = 0; j < kCluster; j += 2u) clusters |= keep.test_pair(keep.cl0 + cluster0 + c0 + j, rch) << j;
                if (c0 + kCluster > n_cl) clusters &= (1u << (n_cl - c0)) - 1u;  // (entries past the table are padding)
            }
            while (clusters != 0u) {
                const uint32_t c = g * kCluster + (uint32_t)__builtin_ctz(clusters);
                clusters &= clusters - 1u;
                uint32_t members = 0u;
                {
                    const float rch = keep.reach();
                    const uint32_t i0 = c * kCluster;
#pragma unroll
                    for (uint32_t j = 0; j < kCluster; j += 2u) members |= keep.test_pair(slot0 + i0 + j, rch) << j;
                    if (i0 + kCluster > n) members &= (1u << (n - i0)) - 1u;
                }
                while (members != 0u) {
                    const uint32_t i = c * kCluster + (uint32_t)__builtin_ctz(members);
                    members &= members - 1u;
                    if (keep.stopped()) return;
                    const Rec rec = recs[i];
                    body(rec, slot0 + i);
                }
            }
        }
    }
}

// The walk for the 64 camera rays of a wave together.  They leave one point and fill a narrow
// pyramid (an 8x8 tile of pixels), and the wave-uniform walk has every lane test every sphere
// that any of them may hit: 64 times the same answer, some 700 times per wave in a scene of
// 10 000 objects.  Here the pyramid is enclosed in a cone (axis a: the mean direction; half-angle:
// the largest angle of a ray to it, with a margin), and the 64 lanes test 64 spheres of a level
// against the CONE at once, as wide_walk_table does against one ray: every group of a table in
// the order nearest first, the 16 clusters of up to four groups the cone touches, the 16 members
// of up to four such clusters.  What is left are the objects near the tile; those are met one
// after the other, each lane with the test of its own ray (may_hit_primary) and the exact
// intersection.  A sphere is ruled out for the whole wave when
//   its centre is farther than r from the cone: with tc = m.a and perp the distance of the centre
//     from the axis, perp cos w - tc sin w is its distance from the line the cone's surface makes
//     in the plane of axis and centre, and the cone lies on the other side of that line;
//   or every ray of the cone points away from it (tc cos w + |m| sin w < 0, camera outside);
//   or it begins farther away (|m| - r) than the largest reach of the wave's rays.
// All three are conservative against the rays' own tests, with allowances for the arithmetic,
// so the objects that reach an exact intersection include all that can be hit: same pixels.
template <typename T, typename Op> RT_DEV T wave_reduce(T v, Op &&op)
{
    for (int off = 32; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off));
    return v;
}
template <typename R>
struct Beam {
    V3<R> a;         // unit axis
    R sin_w, cos_w;  // of the half-angle (NaN: the rays do not fit a cone of less than a right angle, nothing is ruled out by it)
};
// `live`: this lane holds a ray that can hit bounded objects; at least one lane of the wave does
template <typename R> RT_DEV Beam<R> make_beam(V3<R> u, bool live)
{
    auto add = [](R x, R y) { return x + y; };
    auto larger = [](R x, R y) { return Num<R>::max_(x, y); };
    const V3<R> mine = live ? u : V3<R>{ R(0), R(0), R(0) };
    const V3<R> sum = { wave_reduce(mine.x, add), wave_reduce(mine.y, add), wave_reduce(mine.z, add) };
    const R inv = R(1) / Num<R>::sqrt_(dot3(sum, sum));
    Beam<R> b;
    b.a = { sum.x * inv, sum.y * inv, sum.z * inv };
    const R along = dot3(u, b.a);
    const V3<R> w = { cull_fma(-along, b.a.x, u.x), cull_fma(-along, b.a.y, u.y), cull_fma(-along, b.a.z, u.z) };
    const R sin2 = wave_reduce(live ? dot3(w, w) : R(0), larger);
    const bool wide = __ballot(live && !(along > R(0.1))) != 0ull;
    b.sin_w = Num<R>::sqrt_(sin2) * R(1.001) + R(64) * Num<R>::cull_slack;
    if (wide || !(b.sin_w < R(0.99))) b.sin_w = Num<R>::inf - Num<R>::inf;
    b.cos_w = Num<R>::sqrt_(Num<R>::max_(cull_fma(-b.sin_w, b.sin_w, R(1)), R(0)));
    return b;
}
template <typename R>
RT_DEV bool beam_may_hit(const PrimBoundRec<R> &b, const Beam<R> &bm, R reach_max)
{
    const R tc = cull_fma(b.m[2], bm.a.z, cull_fma(b.m[1], bm.a.y, b.m[0] * bm.a.x));
    const R perp = Num<R>::sqrt_(Num<R>::max_(cull_fma(-tc, tc, b.km), R(0)));  // a lower bound
    const R upper = b.dist * (R(1) + R(8) * Num<R>::cull_slack);               // of |m|
    const R e = R(8) * Num<R>::cull_slack * (Num<R>::abs_(tc) + upper);          // rounding of the sums below
    const R miss = (cull_fma(perp, bm.cos_w, -(tc * bm.sin_w)) - e) - b.r;
    const R behind = b.behind - (cull_fma(upper, bm.sin_w, tc * bm.cos_w) + e);
    const R far = (b.dist - b.r) - reach_max;
    return !(Num<R>::max_(Num<R>::max_(miss, behind), far) > R(0));
}
// What the lanes bring to a beam walk, for camera rays: the lane's own ray (unit direction, |direction|,
// closest hit so far).  reach(): how far from the apex this lane still looks (-inf: it does not),
// test(slot): whether its ray may hit the object at `slot`, with the test of that one ray.
template <typename R>
struct CameraLane {
    const PrimBoundRec<R> *bnd;
    V3<R> u;
    R dlen;
    bool live;
    const R *best;
    RT_DEV R reach() const { return live ? *best * dlen : -Num<R>::inf; }
    RT_DEV bool test(uint32_t slot) const { return live & may_hit_primary<R, false>(PrimSphereCull<R, false>::head(bnd + slot), u, *best * dlen); }
};
// For the shadow rays of a wave's hits, which all end at the light: the cone has its apex THERE
// and holds the directions from the light to the hit points; a lane looks as far from the light as
// its hit point lies, until its ray is known to be blocked.
template <typename R>
struct ShadowLane {
    const BoundRec<R> *bnd;
    CullRay<R> cr;  // the shadow ray itself, hit point to light
    R distance;     // of the light along it
    bool live;
    const bool *shadowed;
    RT_DEV bool open() const { return live && !*shadowed; }
    RT_DEV R reach() const { return open() ? distance * cr.dlen : -Num<R>::inf; }
    RT_DEV bool test(uint32_t slot) const { return open() & may_hit<R, false>(load_uniform(bnd + slot), cr, distance * cr.dlen - cr.shift); }
};
// group_order: the table's groups in the order to meet them (table-relative indices) or null for table
// order; from: the spheres relative to the apex.  All 64 lanes call this.
template <typename R, typename Rec, typename Lane, typename Body>
RT_DEV void beam_walk_table(const Rec *recs, uint32_t n, uint32_t slot0, uint32_t cluster0, uint32_t group0, const uint32_t *group_order,
                            const ApexBounds<R> &from, const Beam<R> &beam, const Lane &ray, Body &&body)
{
    static_assert(kCluster == 16, "a quarter of the wave per cluster");
    if (n == 0u) return;
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t quarter = lane >> 4, member = lane & 15u;
    const uint32_t n_cl = (n + kCluster - 1) / kCluster, n_grp = (n_cl + kCluster - 1) / kCluster;
    auto mine = [&](const uint32_t v[4]) { return quarter == 0u ? v[0] : quarter == 1u ? v[1] : quarter == 2u ? v[2] : v[3]; };
    auto wave_reach = [&]() { return wave_reduce(ray.reach(), [](R x, R y) { return Num<R>::max_(x, y); }); };
    R reach_max = wave_reach();
    for (uint32_t g0 = 0; g0 < n_grp; g0 += 64u) {
        if (!(reach_max > -Num<R>::inf)) return;  // no lane is looking any more
        bool hit_g = false;
        uint32_t g_rel = 0u;
        if (g0 + lane < n_grp) {
            g_rel = group_order ? group_order[g0 + lane] : g0 + lane;
            hit_g = beam_may_hit(from.grp[group0 + g_rel], beam, reach_max);
        }
        unsigned long long groups = __ballot(hit_g);
        while (groups != 0ull) {
            uint32_t gpos[4], gidx[4];
            const uint32_t n_g = pop_four(groups, gpos);
            for (int j = 0; j < 4; ++j) gidx[j] = (uint32_t)__builtin_amdgcn_readlane((int)g_rel, (int)gpos[j]);
            const uint32_t c = mine(gidx) * kCluster + member;
            bool hit_c = false;
            if (quarter < n_g && c < n_cl) hit_c = beam_may_hit(from.cl[cluster0 + c], beam, reach_max);
            unsigned long long clusters = __ballot(hit_c);
            while (clusters != 0ull) {
                uint32_t cpos[4], cidx[4];
                const uint32_t n_c = pop_four(clusters, cpos);
                for (int j = 0; j < 4; ++j) cidx[j] = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)cpos[j]);
                const uint32_t i = mine(cidx) * kCluster + member;
                bool hit_o = false;
                if (quarter < n_c && i < n) hit_o = beam_may_hit(from.bnd[slot0 + i], beam, reach_max);
                unsigned long long objects = __ballot(hit_o);
                if (objects == 0ull) continue;
                while (objects != 0ull) {
                    const uint32_t at = (uint32_t)__builtin_ctzll(objects);
                    objects &= objects - 1ull;
                    const uint32_t io = (uint32_t)__builtin_amdgcn_readlane((int)i, (int)at);
                    const bool mine_too = ray.test(slot0 + io);
                    if (__ballot(mine_too) == 0ull) continue;
                    const Rec rec = load_uniform(recs + io);
                    if (mine_too) body(rec, slot0 + io);
                }
                reach_max = wave_reach();
                if (!(reach_max > -Num<R>::inf)) return;
            }
        }
    }
}

// Object::intersect for every object of the world (src/shapes/objects.rs:185-198 with
// src/rays.rs:30-34): the ray goes into object space (direction not renormalised), then the
// per-shape routine runs.  For each object `per(slot, run)` is called once; `run(emit)`
// performs the intersection and calls emit(t) per intersection.
// Objects inside groups: the ray goes through the chain level by level (Group::intersect,
// src/shapes/groups.rs:68-71, then Object::intersect, objects.rs:185-188).
template <typename R, typename Per>
RT_DEV void walk_chained(const SceneDev<R> &S, V3<R> ro, V3<R> rd, Per &&per)
{
    const uint32_t slot0 = S.n - S.n_chn;
    for (uint32_t i = 0; i < S.n_chn; ++i) {
        const ChainRec<R> rec = load_uniform(S.chn + i);
        V3<R> o = ro, d = rd;
        for (uint32_t k = 0; k <= rec.depth && k <= kMaxChain; ++k) {
            const V3<R> o2 = xform_point(rec.m[k], o);
            d = xform_vector(rec.m[k], d);
            o = o2;
        }
        per(slot0 + i, [&](auto &&emit) {
            if (rec.type == SH_SPHERE) {
                isect_sphere(o, d, sphere_c(o), emit);
            } else if (rec.type == SH_PLANE) {
                isect_plane(-o.y, d.y, emit);
            } else if (rec.type == SH_CUBE) {
                const V3<R> lo = { R(-1) - o.x, R(-1) - o.y, R(-1) - o.z };
                const V3<R> hi = { R(1) - o.x, R(1) - o.y, R(1) - o.z };
                isect_cube(lo, hi, d, emit);
            } else if (rec.type == SH_CYLINDER) {
                isect_cylinder(o, twice(o), cylinder_c(o), d, rec.ymin, rec.ymax, rec.closed != 0, emit);
            } else {
                isect_cone(o, twice(o), cone_c(o), d, rec.ymin, rec.ymax, rec.closed != 0, emit);
            }
        });
    }
}

// The five tables and the chained objects with a wave-uniform slot: `bounded(recs, n, slot0, shape,
// body)` walks a table whose objects have finite bounding spheres, `unbounded(...)` one of the other
// two (planes, cones); body(rec, slot) takes the ray into the object's space and intersects.
// UNBOUNDED_FIRST: planes and cones before the rest (what is found there shortens the other walks).
template <typename R, bool CHAIN, bool UNBOUNDED_FIRST, typename Bounded, typename Unbounded, typename Per>
RT_DEV void walk_scene_tables(const SceneDev<R> &S, V3<R> ro, V3<R> rd, Bounded &&bounded, Unbounded &&unbounded, Per &&per)
{
    // the ray in the space of the object at `slot` (wave-uniform branch: the flag is a scalar)
    auto into_object = [&](const R *m, uint32_t slot, V3<R> &o, V3<R> &d) {
        if (ro_load(S.slot_flags + slot) & SLOT_DIAGONAL) {
            o = diag_point(m, ro);
            d = diag_vector(m, rd);
        } else {
            o = xform_point(m, ro);
            d = xform_vector(m, rd);
        }
    };
    auto planes = [&]() {
        unbounded(S.pln, S.n_pln, S.n_sph, (uint32_t)SH_PLANE, [&](const PlaneRec<R> &rec, uint32_t slot) {
            // only the y row of the object-space ray is read by planes.rs:11-17
            R oy, dy;
            if (ro_load(S.slot_flags + slot) & SLOT_DIAGONAL) {
                oy = rec.m[1] * ro.y + rec.m[3];
                dy = rec.m[1] * rd.y;
            } else {
                oy = ((rec.m[0] * ro.x + rec.m[1] * ro.y) + rec.m[2] * ro.z) + rec.m[3];
                dy = (rec.m[0] * rd.x + rec.m[1] * rd.y) + rec.m[2] * rd.z;
            }
            per(slot, [&](auto &&emit) { isect_plane(-oy, dy, emit); });
        });
    };
    auto cones = [&]() {
        unbounded(S.con, S.n_con, S.n_sph + S.n_pln + S.n_cub + S.n_cyl, (uint32_t)SH_CONE, [&](const CappedRec<R> &rec, uint32_t slot) {
            V3<R> o, d;
            into_object(rec.m, slot, o, d);
            per(slot, [&](auto &&emit) { isect_cone(o, twice(o), cone_c(o), d, rec.ymin, rec.ymax, rec.closed != 0, emit); });
        });
    };
    if constexpr (UNBOUNDED_FIRST) { planes(); cones(); }
    bounded(S.sph, S.n_sph, 0u, (uint32_t)SH_SPHERE, [&](const XformRec<R> &rec, uint32_t slot) {
        V3<R> o, d;
        into_object(rec.m, slot, o, d);
        per(slot, [&](auto &&emit) { isect_sphere(o, d, sphere_c(o), emit); });
    });
    if constexpr (!UNBOUNDED_FIRST) planes();
    bounded(S.cub, S.n_cub, S.n_sph + S.n_pln, (uint32_t)SH_CUBE, [&](const XformRec<R> &rec, uint32_t slot) {
        V3<R> o, d;
        into_object(rec.m, slot, o, d);
        const V3<R> lo = { R(-1) - o.x, R(-1) - o.y, R(-1) - o.z };
        const V3<R> hi = { R(1) - o.x, R(1) - o.y, R(1) - o.z };
        per(slot, [&](auto &&emit) { isect_cube(lo, hi, d, emit); });
    });
    bounded(S.cyl, S.n_cyl, S.n_sph + S.n_pln + S.n_cub, (uint32_t)SH_CYLINDER, [&](const CappedRec<R> &rec, uint32_t slot) {
        V3<R> o, d;
        into_object(rec.m, slot, o, d);
        per(slot, [&](auto &&emit) { isect_cylinder(o, twice(o), cylinder_c(o), d, rec.ymin, rec.ymax, rec.closed != 0, emit); });
    });
    if constexpr (!UNBOUNDED_FIRST) cones();
    if constexpr (CHAIN) walk_chained(S, ro, rd, per);
}

// `table(recs, n, slot0, cluster0, group0, body)`: one table, walked wave by wave (keep: SphereCull or
// KeepAll), lane by lane (LaneTreeCull), or by the whole wave for one ray (SphereCull; lane_best, wave_best:
// see wide_walk_table)
template <typename R, bool CHAIN, int WALK = WALK_WAVE, typename Keep, typename Per>
RT_DEV void walk_scene(const SceneDev<R> &S, V3<R> ro, V3<R> rd, const Keep &keep, Per &&per, const R *lane_best = nullptr,
                       R *wave_best = nullptr)
{
    if constexpr ((WALK == WALK_LANE || walk_is_wide(WALK)) && Keep::clustered) {
        uint32_t base = 0;
        auto lane_walk_table = [&](auto recs, uint32_t n, uint32_t slot0, uint32_t cluster0, uint32_t group0, const Keep &k, auto &&body) {
            if constexpr (walk_is_wide(WALK)) wide_walk_table<R>(recs, n, slot0, cluster0, group0, k, lane_best, wave_best, body);
            else lane_tree_walk_table(recs, n, slot0, cluster0, group0, k, body);
        };
        lane_walk_table(S.sph, S.n_sph, base, S.cl_off[SH_SPHERE], S.grp_off[SH_SPHERE], keep_for_shape(keep, SH_SPHERE), [&](const XformRec<R> &rec, uint32_t slot) {
            const V3<R> o = xform_point(rec.m, ro);
            const V3<R> d = xform_vector(rec.m, rd);
            per(slot, [&](auto &&emit) { isect_sphere(o, d, sphere_c(o), emit); });
        });
        base += S.n_sph;
        walk_table(S.pln, S.n_pln, base, 0u, 0u, KeepAll(), [&](const PlaneRec<R> &rec, uint32_t slot) {
            const R oy = ((rec.m[0] * ro.x + rec.m[1] * ro.y) + rec.m[2] * ro.z) + rec.m[3];
            const R dy = (rec.m[0] * rd.x + rec.m[1] * rd.y) + rec.m[2] * rd.z;
            per(slot, [&](auto &&emit) { isect_plane(-oy, dy, emit); });
        });
        base += S.n_pln;
        lane_walk_table(S.cub, S.n_cub, base, S.cl_off[SH_CUBE], S.grp_off[SH_CUBE], keep_for_shape(keep, SH_CUBE), [&](const XformRec<R> &rec, uint32_t slot) {
            const V3<R> o = xform_point(rec.m, ro);
            const V3<R> d = xform_vector(rec.m, rd);
            const V3<R> lo = { R(-1) - o.x, R(-1) - o.y, R(-1) - o.z };
            const V3<R> hi = { R(1) - o.x, R(1) - o.y, R(1) - o.z };
            per(slot, [&](auto &&emit) { isect_cube(lo, hi, d, emit); });
        });
        base += S.n_cub;
        lane_walk_table(S.cyl, S.n_cyl, base, S.cl_off[SH_CYLINDER], S.grp_off[SH_CYLINDER], keep, [&](const CappedRec<R> &rec, uint32_t slot) {
            const V3<R> o = xform_point(rec.m, ro);
            const V3<R> d = xform_vector(rec.m, rd);
            per(slot, [&](auto &&emit) { isect_cylinder(o, twice(o), cylinder_c(o), d, rec.ymin, rec.ymax, rec.closed != 0, emit); });
        });
        base += S.n_cyl;
        walk_table(S.con, S.n_con, base, 0u, 0u, KeepAll(), [&](const CappedRec<R> &rec, uint32_t slot) {
            const V3<R> o = xform_point(rec.m, ro);
            const V3<R> d = xform_vector(rec.m, rd);
            per(slot, [&](auto &&emit) { isect_cone(o, twice(o), cone_c(o), d, rec.ymin, rec.ymax, rec.closed != 0, emit); });
        });
        if constexpr (CHAIN) walk_chained(S, ro, rd, per);
    } else {
        walk_scene_tables<R, CHAIN, false>(S, ro, rd, [&](auto recs, uint32_t n, uint32_t slot0, uint32_t shape, auto &&body) {
            walk_table(recs, n, slot0, S.cl_off[shape], S.grp_off[shape], keep_for_shape(keep, shape), body);
        }, [&](auto recs, uint32_t n, uint32_t slot0, uint32_t shape, auto &&body) {
            walk_table(recs, n, slot0, 0u, 0u, KeepAll(), body);
        }, per);
    }
}

// The same walk for primary rays: they all start at the camera, so the object-space origin
// and every term derived from it alone were evaluated once per object (k_primary_tables,
// with the expressions above) and are read here as wave-uniform scalars.
template <typename R> RT_DEV V3<R> xform_vector9(const R *m, V3<R> v)
{
    return { (m[0] * v.x + m[1] * v.y) + m[2] * v.z,
             (m[3] * v.x + m[4] * v.y) + m[5] * v.z,
             (m[6] * v.x + m[7] * v.y) + m[8] * v.z };
}
// a direction into the space of the object at `slot` (diag_vector when its 3x3 is diagonal)
template <typename R> RT_DEV V3<R> primary_direction(const SceneDev<R> &S, uint32_t slot, const R *m, V3<R> v)
{
    if (ro_load(S.slot_flags + slot) & SLOT_DIAGONAL) return { m[0] * v.x, m[4] * v.y, m[8] * v.z };
    return xform_vector9(m, v);
}
// `bounded(recs, n, slot0, shape, body)` walks one of the three tables whose objects have finite
// bounding spheres; `unbounded(run)` runs the walks of the others (for the lanes that hold a ray).
template <typename R, bool CHAIN, typename Bounded, typename Unbounded, typename Per>
RT_DEV void walk_primary_tables(const SceneDev<R> &S, const PrimScene<R> &P, V3<R> ro, V3<R> rd, Bounded &&bounded,
                                Unbounded &&unbounded, Per &&per)
{
    // planes and cones first: what a ray hits there shortens its reach for the tables with a hierarchy
    unbounded([&]() {
        walk_table(P.pln, P.n_pln, P.n_sph, 0u, 0u, KeepAll(), [&](const PrimPlaneRec<R> &rec, uint32_t slot) {
            const R dy = (ro_load(S.slot_flags + slot) & SLOT_DIAGONAL) ? rec.m[1] * rd.y : (rec.m[0] * rd.x + rec.m[1] * rd.y) + rec.m[2] * rd.z;
            per(slot, [&](auto &&emit) { isect_plane(rec.neg_oy, dy, emit); });
        });
        walk_table(P.con, P.n_con, P.n_sph + P.n_pln + P.n_cub + P.n_cyl, 0u, 0u, KeepAll(), [&](const PrimCappedRec<R> &rec, uint32_t slot) {
            const V3<R> o = { rec.o[0], rec.o[1], rec.o[2] };
            const V3<R> o2 = { rec.o2[0], rec.o2[1], rec.o2[2] };
            const V3<R> d = primary_direction(S, slot, rec.m, rd);
            per(slot, [&](auto &&emit) { isect_cone(o, o2, rec.c, d, rec.ymin, rec.ymax, rec.closed != 0, emit); });
        });
        if constexpr (CHAIN) walk_chained(S, ro, rd, per);
    });
    bounded(P.sph, P.n_sph, 0u, (uint32_t)SH_SPHERE, [&](const PrimSphereRec<R> &rec, uint32_t slot) {
        const V3<R> o = { rec.o[0], rec.o[1], rec.o[2] };
        const V3<R> d = primary_direction(S, slot, rec.m, rd);
        per(slot, [&](auto &&emit) { isect_sphere(o, d, rec.c, emit); });
    });
    bounded(P.cub, P.n_cub, P.n_sph + P.n_pln, (uint32_t)SH_CUBE, [&](const PrimCubeRec<R> &rec, uint32_t slot) {
        const V3<R> d = primary_direction(S, slot, rec.m, rd);
        const V3<R> lo = { rec.lo[0], rec.lo[1], rec.lo[2] };
        const V3<R> hi = { rec.hi[0], rec.hi[1], rec.hi[2] };
        per(slot, [&](auto &&emit) { isect_cube(lo, hi, d, emit); });
    });
    bounded(P.cyl, P.n_cyl, P.n_sph + P.n_pln + P.n_cub, (uint32_t)SH_CYLINDER, [&](const PrimCappedRec<R> &rec, uint32_t slot) {
        const V3<R> o = { rec.o[0], rec.o[1], rec.o[2] };
        const V3<R> o2 = { rec.o2[0], rec.o2[1], rec.o2[2] };
        const V3<R> d = primary_direction(S, slot, rec.m, rd);
        per(slot, [&](auto &&emit) { isect_cylinder(o, o2, rec.c, d, rec.ymin, rec.ymax, rec.closed != 0, emit); });
    });
}
template <typename R, bool CHAIN, typename Keep, typename Per>
RT_DEV void walk_scene_primary(const SceneDev<R> &S, const PrimScene<R> &P, V3<R> ro, V3<R> rd, const Keep &keep, Per &&per)
{
    const uint32_t *const order = Keep::clustered ? P.grp_order : nullptr;
    walk_primary_tables<R, CHAIN>(S, P, ro, rd, [&](auto recs, uint32_t n, uint32_t slot0, uint32_t shape, auto &&body) {
        walk_table(recs, n, slot0, S.cl_off[shape], S.grp_off[shape], keep, body, order ? order + S.grp_off[shape] : nullptr);
    }, [&](auto &&run) { run(); }, per);
}
// The camera rays of a whole wave (beam_walk_table); every lane calls this, `valid`: it holds a ray.
template <typename R, bool CHAIN, typename Per>
RT_DEV void walk_scene_primary_beam(const SceneDev<R> &S, const PrimScene<R> &P, V3<R> ro, V3<R> rd, bool valid, const R *best, Per &&per)
{
    const CullRay<R> cr = cull_ray<R, false>(S, ro, rd);
    const CameraLane<R> ray = { P.eye.bnd, cr.u, cr.dlen, valid && !cr.dead, best };
    const bool any = __ballot(ray.live) != 0ull;
    Beam<R> beam = {};
    if (any) beam = make_beam(cr.u, ray.live);
    walk_primary_tables<R, CHAIN>(S, P, ro, rd, [&](auto recs, uint32_t n, uint32_t slot0, uint32_t shape, auto &&body) {
        if (any) beam_walk_table(recs, n, slot0, S.cl_off[shape], S.grp_off[shape], P.grp_order + S.grp_off[shape], P.eye, beam, ray, body);
    }, [&](auto &&run) { if (valid) run(); }, per);
}

// Fill the primary-ray table entry of one slot from the scene tables and the camera origin.
template <typename R, typename Rec> RT_DEV void copy3x3(const R *m, Rec &out)
{
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) out.m[r * 3 + k] = m[r * 4 + k];
}
template <typename R>
RT_DEV void make_primary_record(const SceneDev<R> &S, const PrimScene<R> &P, V3<R> cam_origin, uint32_t slot)
{
    uint32_t i = slot;
    if (i < S.n_sph) {
        PrimSphereRec<R> out;
        copy3x3(S.sph[i].m, out);
        const V3<R> o = xform_point(S.sph[i].m, cam_origin);
        out.o[0] = o.x; out.o[1] = o.y; out.o[2] = o.z;
        out.c = sphere_c(o);
        out.pad = R(0);
        const_cast<PrimSphereRec<R> *>(P.sph)[i] = out;
        return;
    }
    i -= S.n_sph;
    if (i < S.n_pln) {
        PrimPlaneRec<R> out;
        const R *m = S.pln[i].m;
        out.m[0] = m[0]; out.m[1] = m[1]; out.m[2] = m[2];
        out.neg_oy = -(((m[0] * cam_origin.x + m[1] * cam_origin.y) + m[2] * cam_origin.z) + m[3]);
        const_cast<PrimPlaneRec<R> *>(P.pln)[i] = out;
        return;
    }
    i -= S.n_pln;
    if (i < S.n_cub) {
        PrimCubeRec<R> out;
        copy3x3(S.cub[i].m, out);
        const V3<R> o = xform_point(S.cub[i].m, cam_origin);
        out.lo[0] = R(-1) - o.x; out.lo[1] = R(-1) - o.y; out.lo[2] = R(-1) - o.z;
        out.hi[0] = R(1) - o.x; out.hi[1] = R(1) - o.y; out.hi[2] = R(1) - o.z;
        out.pad = R(0);
        const_cast<PrimCubeRec<R> *>(P.cub)[i] = out;
        return;
    }
    i -= S.n_cub;
    const bool is_cyl = i < S.n_cyl;
    if (!is_cyl) i -= S.n_cyl;
    if (!is_cyl && i >= S.n_con) return;
    const CappedRec<R> &src = is_cyl ? S.cyl[i] : S.con[i];
    PrimCappedRec<R> out;
    copy3x3(src.m, out);
    const V3<R> o = xform_point(src.m, cam_origin);
    const V3<R> o2 = twice(o);
    out.o[0] = o.x; out.o[1] = o.y; out.o[2] = o.z;
    out.o2[0] = o2.x; out.o2[1] = o2.y; out.o2[2] = o2.z;
    out.c = is_cyl ? cylinder_c(o) : cone_c(o);
    out.ymin = src.ymin; out.ymax = src.ymax; out.closed = src.closed;
    for (auto &w : out.pad) w = 0;
    const_cast<PrimCappedRec<R> *>(is_cyl ? P.cyl : P.con)[i] = out;
}

// Place of group `g` (index into S.grp) among the groups of its shape table when they are ordered by
// how far along a ray from the camera they can first be met (distance of the centre minus the
// radius; unbounded groups first, ties by index): one thread per group, a count over its table.
template <typename R>
RT_DEV void rank_group(const SceneDev<R> &S, const PrimScene<R> &P, V3<R> cam_origin, uint32_t g)
{
    const uint32_t counts[5] = { S.n_sph, S.n_pln, S.n_cub, S.n_cyl, S.n_con };
    for (uint32_t shape = 0; shape < 5u; ++shape) {
        if (shape == SH_PLANE || shape == SH_CONE) continue;  // never walked through the hierarchy
        const uint32_t n_cl = (counts[shape] + kCluster - 1) / kCluster, n_grp = (n_cl + kCluster - 1) / kCluster;
        const uint32_t first = S.grp_off[shape];
        if (g < first || g >= first + n_grp) continue;
        auto key = [&](uint32_t k) {
            const BoundRec<R> b = S.grp[k];
            const V3<R> m = { b.c[0] - cam_origin.x, b.c[1] - cam_origin.y, b.c[2] - cam_origin.z };
            const R d = Num<R>::sqrt_(dot3(m, m)) - b.r;
            return d == d ? d : -Num<R>::inf;  // an unbounded group (inf - inf) is met first
        };
        const R mine = key(g);
        uint32_t rank = 0;
        for (uint32_t k = first; k < first + n_grp; ++k) {
            const R other = key(k);
            rank += (other < mine || (other == mine && k < g)) ? 1u : 0u;
        }
        const_cast<uint32_t *>(P.grp_order)[first + rank] = g - first;
        return;
    }
}

// Where a kernel reads the shading records from.  The rays of a bounce wave hit different
// objects, so the record of a hit is gathered per lane, 424 bytes (f64) from 64 places: through
// the vector cache that is some 1700 separate accesses per wave and node, and the cache's
// address unit, not the arithmetic, sets the pace.  Kernels instantiated with COLD_LDS copy the
// table of a small scene into LDS when a workgroup starts (stage_cold_table) and gather from
// there with LDS reads; the others read the table in memory.
extern __shared__ __attribute__((aligned(16))) unsigned char rtx_lds_dyn[];
constexpr uint32_t kColdLdsObjects = 32;  // largest scene whose shading table a workgroup keeps in LDS
// The copy holds the shading records, then the slot -> object index table (as many entries).
template <typename R> __host__ __device__ inline uint32_t cold_lds_bytes(uint32_t n_objects)
{
    return n_objects <= kColdLdsObjects ? n_objects * (uint32_t)(sizeof(ColdRec<R>) + sizeof(uint32_t)) : 0u;
}
template <typename R, bool COLD_LDS> struct ColdTable {
    using Rec = const ColdRec<R>;
    using Index = const uint32_t;
    static RT_DEV Rec *base(const SceneDev<R> &S) { return S.cold; }
    static RT_DEV Index *slot_obj(const SceneDev<R> &S) { return S.slot_obj; }
};
template <typename R> struct ColdTable<R, true> {
    using Rec = const __attribute__((address_space(3))) ColdRec<R>;
    using Index = const __attribute__((address_space(3))) uint32_t;
    static RT_DEV Rec *base(const SceneDev<R> &) { return (Rec *)rtx_lds_dyn; }
    static RT_DEV Index *slot_obj(const SceneDev<R> &S) { return (Index *)(rtx_lds_dyn + S.n * (uint32_t)sizeof(ColdRec<R>)); }
};
template <typename R> RT_DEV void stage_cold_table(const SceneDev<R> &S, uint32_t n_threads)
{
    const uint4 *from = reinterpret_cast<const uint4 *>(S.cold);
    uint4 *to = reinterpret_cast<uint4 *>(rtx_lds_dyn);
    const uint32_t n16 = S.n * (uint32_t)sizeof(ColdRec<R>) / 16u;
    for (uint32_t i = threadIdx.x; i < n16; i += n_threads) to[i] = from[i];
    uint32_t *index = reinterpret_cast<uint32_t *>(rtx_lds_dyn + S.n * (uint32_t)sizeof(ColdRec<R>));
    for (uint32_t i = threadIdx.x; i < S.n; i += n_threads) index[i] = S.slot_obj[i];
    __syncthreads();
}

// ---- closest hit over the whole world ---------------------------------------------------
// The reference concatenates every object's intersections, stable-sorts by t and takes the
// first t >= 0 (src/scenes/world.rs:98-119): the smallest t >= 0, and among equal t the object
// that comes first in World.objects.  The tables are walked shape by shape, so a tie between
// two objects is settled by looking up their World indices (rare path).
//
// With TRACK the same pass also derives n1/n2 of prepare_computations
// (world.rs:253-285).  Everything that precedes the hit in the sorted list has t < 0.  An
// object sits in `containers` at the hit iff it has an odd number of negative t's, and the
// list is ordered by each object's largest negative t (ties: World order).  Only the last
// entry and, when the hit object itself is that entry, the one before it are ever read, so
// the two "latest" open objects are all that must be remembered.
template <typename R>
struct Hit {
    R t;
    uint32_t slot;     // 0xffffffff = miss
    uint32_t obj;      // index in World.objects (valid when slot is)
    bool hit_open;     // TRACK: the hit object is in `containers` before it is toggled
    R top1_t, top2_t;  // TRACK: largest negative t of the last / second-to-last container
    uint32_t top1, top2;  // TRACK: their slots
};

constexpr uint32_t kNone = 0xffffffffu;

// An object with an odd number of intersections behind the ray's origin is in `containers` when the
// hit is reached; `last_neg` (its largest negative t) places it in that list.  Keeps the last two.
template <typename R, typename Before>
RT_DEV void note_open_container(Hit<R> &h, uint32_t slot, R last_neg, Before &&before)
{
    // later in the sorted list = larger t, or equal t and later in World.objects
    const bool after1 = last_neg > h.top1_t || (last_neg == h.top1_t && (h.top1 == kNone || before(h.top1, slot)));
    if (after1) {
        h.top2 = h.top1; h.top2_t = h.top1_t;
        h.top1 = slot; h.top1_t = last_neg;
    } else {
        const bool after2 = last_neg > h.top2_t || (last_neg == h.top2_t && (h.top2 == kNone || before(h.top2, slot)));
        if (after2) { h.top2 = slot; h.top2_t = last_neg; }
    }
}

template <typename R, bool TRACK, bool COLD_LDS = false, bool WIDE = false, typename Walk>
RT_DEV Hit<R> closest_hit_with(const SceneDev<R> &S, Walk &&walk)
{
    Hit<R> h;
    h.t = Num<R>::inf;
    h.slot = kNone;
    h.obj = kNone;
    h.hit_open = false;
    h.top1 = h.top2 = kNone;
    h.top1_t = h.top2_t = -Num<R>::inf;
    // true when `slot` precedes `other` in World.objects
    auto before = [&](uint32_t slot, uint32_t other) { return ro_load(S.slot_obj + slot) < ro_load(S.slot_obj + other); };
    walk(&h.t, [&](uint32_t slot, auto &&run) {
        if constexpr (!TRACK) {
            run([&](R t, bool ok = true) {
                if (ok && t >= R(0) && t <= h.t) {
                    if (t < h.t || h.slot == kNone || (h.slot != slot && before(slot, h.slot))) { h.t = t; h.slot = slot; }
                }
            });
        } else {
            R best = Num<R>::inf;       // smallest t >= 0 of this object
            R last_neg = -Num<R>::inf;  // largest t < 0 of this object
            uint32_t n_neg = 0;
            bool any_pos = false;
            run([&](R t, bool ok = true) {
                const bool pos = ok && t >= R(0), neg = ok && t < R(0);
                best = Num<R>::min_(best, pos ? t : Num<R>::inf);
                any_pos = any_pos || pos;
                n_neg += neg ? 1u : 0u;
                last_neg = Num<R>::max_(last_neg, neg ? t : -Num<R>::inf);
            });
            bool open = false;
            if (n_neg & 1u) open = (ro_load(&S.cold[ro_load(S.slot_obj + slot)].flags) & OF_CONTAINER) != 0;
            if (any_pos && best <= h.t) {
                if (best < h.t || h.slot == kNone || before(slot, h.slot)) { h.t = best; h.slot = slot; h.hit_open = open; }
            }
            if (open) note_open_container(h, slot, last_neg, before);
        }
    });
    if (h.slot != kNone) {
        if constexpr (COLD_LDS) h.obj = ColdTable<R, true>::slot_obj(S)[h.slot];
        else h.obj = ro_load(S.slot_obj + h.slot);
    }
    if constexpr (WIDE) {
        // one ray, every lane with the closest hit among the objects it tested itself: the ray's is the
        // smallest t, and among equal t the object that comes first in World.objects
        static_assert(!TRACK, "the container bookkeeping is not joined across lanes");
        const R t_min = wave_min(h.t);
        const uint32_t mine = (h.slot != kNone && h.t == t_min) ? h.obj : kNone;
        const uint32_t first = wave_min(mine);
        const unsigned long long who = __ballot(mine == first && first != kNone);
        if (who == 0ull) {
            h.t = Num<R>::inf;
            h.slot = h.obj = kNone;
        } else {
            const int src = __builtin_ctzll(who);
            h.t = __shfl(h.t, src);
            h.slot = __shfl(h.slot, src);
            h.obj = first;
        }
    }
    return h;
}
// The closest-hit walks.  t_limit: hits beyond it need not be found (inf for color_at; the
// distance to the light for a shadow ray, which only asks whether the closest hit lies before it).
template <typename R, bool TRACK, bool CULL, bool CHAIN, int WALK = WALK_WAVE, bool COLD_LDS = false>
RT_DEV Hit<R> closest_hit(const SceneDev<R> &S, V3<R> ro, V3<R> rd, R t_limit)
{
    if constexpr (CULL && WALK == WALK_TREE_WIDE) {
        static_assert(!TRACK, "the packed hierarchy is walked ahead of the origin only; containers come from containers_behind");
        return closest_hit_with<R, TRACK, COLD_LDS, true>(S, [&](const R *best, auto &&per) {
            R wave_best = Num<R>::inf;
            const LaneTreeCull<R> keep = lane_tree_cull<R>(S, ro, rd, &wave_best, t_limit, nullptr);
            walk_scene<R, CHAIN, WALK_TREE_WIDE>(S, ro, rd, keep, per, best, &wave_best);
        });
    } else if constexpr (CULL && walk_in_lds(WALK)) {
        static_assert(!TRACK, "the packed hierarchy is walked ahead of the origin only; containers come from containers_behind");
        return closest_hit_with<R, TRACK, COLD_LDS>(S, [&](const R *best, auto &&per) {
            const LaneTreeCull<R> keep = lane_tree_cull<R>(S, ro, rd, best, t_limit, nullptr);
            walk_scene<R, CHAIN, WALK == WALK_LANE ? WALK_LANE : WALK_WAVE>(S, ro, rd, keep, per);
        });
    } else if constexpr (CULL && WALK == WALK_WIDE) {
        return closest_hit_with<R, TRACK, COLD_LDS, true>(S, [&](const R *best, auto &&per) {
            R wave_best = Num<R>::inf;  // the closest hit any lane has found so far (refreshed after every batch of exact tests)
            const SphereCull<R, TRACK> keep = { S.bnd, S.cl, S.grp, cull_ray<R, TRACK>(S, ro, rd), &wave_best, t_limit, nullptr };
            walk_scene<R, CHAIN, WALK_WIDE>(S, ro, rd, keep, per, best, &wave_best);
        });
    } else if constexpr (CULL) {
        return closest_hit_with<R, TRACK, COLD_LDS>(S, [&](const R *best, auto &&per) {
            const SphereCull<R, TRACK> keep = { S.bnd, S.cl, S.grp, cull_ray<R, TRACK>(S, ro, rd), best, t_limit, nullptr };
            walk_scene<R, CHAIN, WALK>(S, ro, rd, keep, per);
        });
    } else {
        return closest_hit_with<R, TRACK, COLD_LDS>(S, [&](const R *, auto &&per) { walk_scene<R, CHAIN>(S, ro, rd, KeepAll(), per); });
    }
}
template <typename R, bool TRACK, bool CULL, bool CHAIN>
RT_DEV Hit<R> closest_hit_primary(const SceneDev<R> &S, const PrimScene<R> &P, V3<R> ro, V3<R> rd)
{
    if constexpr (CULL) {
        return closest_hit_with<R, TRACK>(S, [&](const R *best, auto &&per) {
            const CullRay<R> cr = cull_ray<R, TRACK>(S, ro, rd);
            const PrimSphereCull<R, TRACK> keep = { P.eye.bnd, P.eye.cl, P.eye.grp, cr.u, cr.dlen, cr.dead, best };
            walk_scene_primary<R, CHAIN>(S, P, ro, rd, keep, per);
        });
    } else {
        return closest_hit_with<R, TRACK>(S, [&](const R *, auto &&per) { walk_scene_primary<R, CHAIN>(S, P, ro, rd, KeepAll(), per); });
    }
}

// The same for the 64 camera rays of a wave at once, in scenes with the sphere hierarchy and no
// container bookkeeping.  Called by every lane of the wave; `valid`: this lane holds a ray.
template <typename R, bool CHAIN>
RT_DEV Hit<R> closest_hit_primary_beam(const SceneDev<R> &S, const PrimScene<R> &P, V3<R> ro, V3<R> rd, bool valid)
{
    return closest_hit_with<R, false>(S, [&](const R *best, auto &&per) { walk_scene_primary_beam<R, CHAIN>(S, P, ro, rd, valid, best, per); });
}

// The container bookkeeping of prepare_computations (world.rs:253-285) as a walk of its own, for a ray
// whose closest hit `h` is known.  n1 and n2 are read by refracted_color and schlick alone, both
// behind `transparency != 0`, so in a scene of thousands of objects of which a few are glass almost
// no ray needs them -- while finding them in the same pass as the hit means looking at every
// intersection on the whole line (the objects behind the origin as well), which rules out everything
// that makes the walk over the sphere hierarchy fast (nothing behind the origin, nothing beyond the
// closest hit so far, whole waves at a time).  What precedes the hit in the reference's sorted list are
// the intersections with t < 0: exactly what the ray sent backwards from the same origin meets.  This
// walk therefore culls with that reversed ray (unlimited reach) and intersects the objects it is left
// with exactly as the forward walk does (same ray, same t), keeping the negative t only.
template <typename R, bool CHAIN, int WALK = WALK_WAVE>
RT_DEV void containers_behind(const SceneDev<R> &S, V3<R> ro, V3<R> rd, Hit<R> &h)
{
    auto before = [&](uint32_t slot, uint32_t other) { return ro_load(S.slot_obj + slot) < ro_load(S.slot_obj + other); };
    h.hit_open = false;
    h.top1 = h.top2 = kNone;
    h.top1_t = h.top2_t = -Num<R>::inf;
    const V3<R> back = { -rd.x, -rd.y, -rd.z };
    auto per = [&](uint32_t slot, auto &&run) {
        R last_neg = -Num<R>::inf;
        uint32_t n_neg = 0;
        run([&](R t, bool ok = true) {
            const bool neg = ok && t < R(0);
            n_neg += neg ? 1u : 0u;
            last_neg = Num<R>::max_(last_neg, neg ? t : -Num<R>::inf);
        });
        bool open = false;
        if (n_neg & 1u) open = (ro_load(&S.cold[ro_load(S.slot_obj + slot)].flags) & OF_CONTAINER) != 0;
        if (slot == h.slot) h.hit_open = open;
        if (open) note_open_container(h, slot, last_neg, before);
    };
    if constexpr (walk_in_lds(WALK)) {  // (a wave that holds one ray walks it as a wave of 64 equal rays)
        // A sphere's two roots and a cube's two slab distances have an odd number of negative values only when
        // they straddle zero, which is the origin inside the shape: of those two tables the objects whose
        // bounding sphere holds the origin are enough (reach 0).  The other shapes can be crossed an odd number
        // of times behind the origin (open ends, the caps' own tests) and are looked at all the way back.
        LaneTreeCull<R> keep = lane_tree_cull<R>(S, ro, back, nullptr, Num<R>::inf, nullptr);
        keep.limit_closed = R(0);
        walk_scene<R, CHAIN, WALK == WALK_LANE ? WALK_LANE : WALK_WAVE>(S, ro, rd, keep, per);
    } else {
        const SphereCull<R, false> keep = { S.bnd, S.cl, S.grp, cull_ray<R, false>(S, ro, back), nullptr, Num<R>::inf, nullptr };
        walk_scene<R, CHAIN, WALK_WAVE>(S, ro, rd, keep, per);
    }
}

// src/scenes/world.rs:146-158.  shadowed iff the closest hit (t >= 0) lies before the light
// and its object casts a shadow.  When every object casts, any hit with 0 <= t < distance
// decides it, so the walk needs no closest-hit bookkeeping.
template <typename R, uint32_t SF, int WALK = WALK_WAVE, bool COLD_LDS = false>
RT_DEV bool is_shadowed(const SceneDev<R> &S, V3<R> p, V3<R> dir, R distance)
{
    constexpr bool CULL = (SF & SF_CULL) != 0;
    constexpr bool CHAIN = (SF & SF_CHAIN) != 0;
    if constexpr ((SF & SF_ALL_CAST) != 0) {
        bool shadowed = false;
        auto per = [&](uint32_t, auto &&run) {
            run([&](R t, bool ok = true) {
                if (ok && t >= R(0) && t < distance) shadowed = true;
            });
        };
        if constexpr (CULL && walk_in_lds(WALK)) {
            const LaneTreeCull<R> keep = lane_tree_cull<R>(S, p, dir, nullptr, distance, &shadowed);
            walk_scene<R, CHAIN, WALK>(S, p, dir, keep, per);
            if constexpr (WALK == WALK_TREE_WIDE) shadowed = __ballot(shadowed) != 0ull;  // any lane's object
        } else if constexpr (CULL) {
            const SphereCull<R, false> keep = { S.bnd, S.cl, S.grp, cull_ray<R, false>(S, p, dir), nullptr, distance, &shadowed };
            walk_scene<R, CHAIN, WALK>(S, p, dir, keep, per);
            if constexpr (WALK == WALK_WIDE) shadowed = __ballot(shadowed) != 0ull;  // any lane's object
        } else {
            walk_scene<R, CHAIN>(S, p, dir, KeepAll(), per);
        }
        return shadowed;
    } else {
        Hit<R> h = closest_hit<R, false, CULL, CHAIN, WALK, COLD_LDS>(S, p, dir, distance);
        if (h.slot == kNone) return false;
        return h.t < distance && (ColdTable<R, COLD_LDS>::base(S)[h.obj].flags & OF_CASTS_SHADOW) != 0;
    }
}

// is_shadowed for the hits of a wave of camera rays, in scenes with the sphere hierarchy in which
// every object casts a shadow: their shadow rays end at one point, the light, so the wave walks the
// hierarchy once with a cone from the light around the hit points (beam_walk_table, ShadowLane).
// The rays tested exactly are the reference's, hit point to light.  Called by every lane;
// `asks`: this lane has a hit to shade.
template <typename R, uint32_t SF>
RT_DEV bool is_shadowed_beam(const SceneDev<R> &S, const PrimScene<R> &P, V3<R> p, V3<R> dir, R distance, bool asks)
{
    static_assert((SF & SF_ALL_CAST) != 0 && (SF & SF_CULL) != 0, "any-hit shadows over the hierarchy");
    constexpr bool CHAIN = (SF & SF_CHAIN) != 0;
    bool shadowed = false;
    auto per = [&](uint32_t, auto &&run) {
        run([&](R t, bool ok = true) {
            if (ok && t >= R(0) && t < distance) shadowed = true;
        });
    };
    const CullRay<R> cr = cull_ray<R, false>(S, p, dir);
    const ShadowLane<R> ray = { S.bnd, cr, distance, asks && !cr.dead, &shadowed };
    Beam<R> beam = {};
    const bool any = __ballot(ray.live) != 0ull;
    if (any) beam = make_beam<R>({ -cr.u.x, -cr.u.y, -cr.u.z }, ray.live);
    if (any && !(beam.sin_w < P.shadow_beam_sin)) {
        // Hit points spread in depth (a tile across the edge of an object) are far apart as the light
        // sees them; the cone then holds much of the scene and the rays are better off on their own.
        if (asks) {
            const SphereCull<R, false> keep = { S.bnd, S.cl, S.grp, cr, nullptr, distance, &shadowed };
            walk_scene<R, CHAIN, WALK_WAVE>(S, p, dir, keep, per);
        }
        return shadowed;
    }
    walk_scene_tables<R, CHAIN, true>(S, p, dir, [&](auto recs, uint32_t n, uint32_t slot0, uint32_t shape, auto &&body) {
        if (any) beam_walk_table(recs, n, slot0, S.cl_off[shape], S.grp_off[shape], (const uint32_t *)nullptr, P.light, beam, ray, body);
    }, [&](auto recs, uint32_t n, uint32_t slot0, uint32_t, auto &&body) {
        if (asks && !shadowed) walk_table(recs, n, slot0, 0u, 0u, KeepAll(), body);
    }, [&](uint32_t slot, auto &&run) { if (asks) per(slot, run); });
    return shadowed;
}

// per-shape normals: spheres.rs:20-22, planes.rs:7-9, cubes.rs:53-63, cylinders.rs:96-105,
// cones.rs:114-127
template <typename R>
RT_DEV V3<R> local_normal(uint32_t type, R ymin, R ymax, V3<R> p)
{
    if (type == SH_SPHERE) return p;
    if (type == SH_PLANE) return { R(0), R(1), R(0) };
    if (type == SH_CUBE) {
        R ax = Num<R>::abs_(p.x), ay = Num<R>::abs_(p.y), az = Num<R>::abs_(p.z);
        R maxc = Num<R>::max_(Num<R>::max_(ax, ay), az);
        if (maxc == ax) return { p.x, R(0), R(0) };
        if (maxc == ay) return { R(0), p.y, R(0) };
        return { R(0), R(0), p.z };
    }
    R d2 = p.x * p.x + p.z * p.z;
    if (d2 < R(1) && p.y >= ymax - Num<R>::eps) return { R(0), R(1), R(0) };
    if (d2 < R(1) && p.y <= ymin + Num<R>::eps) return { R(0), R(-1), R(0) };
    if (type == SH_CYLINDER) return { p.x, R(0), p.z };
    R y = Num<R>::sqrt_(d2);
    if (p.y > R(0)) y = -y;
    return { p.x, y, p.z };
}

// Rust `as isize` parity of floor(x): saturating cast, NaN -> 0 (src/surfaces/patterns.rs:145,167).
// Saturated values (|x| >= 2^63) are even except isize::MAX, which is odd.
template <typename R> RT_DEV bool floor_is_even_isize(R f)
{
    if (!(f == f)) return true;
    if (f >= R(9223372036854775807.0)) return false;  // isize::MAX is odd
    if (f <= R(-9223372036854775808.0)) return true;
    return (((long long)f) & 1LL) == 0;
}
// Rust `as usize` parity (patterns.rs:159): negatives and NaN -> 0, saturation at usize::MAX (odd)
template <typename R> RT_DEV bool floor_is_even_usize(R f)
{
    if (!(f == f) || f <= R(0)) return true;
    if (f >= R(18446744073709551615.0)) return false;
    return (((unsigned long long)f) & 1ULL) == 0;
}

// src/surfaces/patterns.rs:94-103, 144-172 on a pattern-space point
template <typename R>
RT_DEV V3<R> pattern_at(uint32_t type, V3<R> c1, V3<R> c2, V3<R> p)
{
    if (type == PT_STRIPE) return floor_is_even_isize(Num<R>::floor_(p.x)) ? c1 : c2;
    if (type == PT_GRADIENT) {
        R fr = p.x - Num<R>::floor_(p.x);
        return { c1.x + (c2.x - c1.x) * fr, c1.y + (c2.y - c1.y) * fr, c1.z + (c2.z - c1.z) * fr };
    }
    if (type == PT_RING) return floor_is_even_usize(Num<R>::floor_(Num<R>::sqrt_(p.x * p.x + p.z * p.z))) ? c1 : c2;
    if (type == PT_CHECKER)
        return floor_is_even_isize((Num<R>::floor_(p.x) + Num<R>::floor_(p.y)) + Num<R>::floor_(p.z)) ? c1 : c2;
    if (type == PT_TEST) return p;
    return { R(0), R(0), R(0) };
}

// ---- one color_at node -------------------------------------------------------------------
// Everything World::color_at does for one ray except the two recursive calls, which come
// back as child rays with the scalar factor their colour is multiplied by on the way up
// (src/scenes/world.rs:121-190, 227-287).
template <typename R>
struct Node {
    bool hit;
    uint32_t obj;
    R t;
    V3<R> point, normalv, over_point, under_point, reflectv;
    bool inside, shadowed;
    R n1, n2;
    V3<R> lighting;       // lighting() term of shade_hit
    bool has_reflect;     // reflected_color recurses: Ray(over_point, reflectv)
    R reflect_factor;     // reflective [* schlick]
    bool has_refract;     // refracted_color recurses: Ray(under_point, refract_dir)
    V3<R> refract_dir;
    R refract_factor;     // transparency [* (1 - schlick)]
};

// `children(nd)` is called for every ray that hits something, as soon as its child rays are
// known and before the shadow ray and the lighting are evaluated: a kernel that queues the
// children there has their registers free during the second walk.  The call sits in divergent
// control flow (the lanes whose ray missed are not there): wave-level operations inside it see
// the lanes that hit.
struct NoChildrenHook {
    template <typename N> RT_DEV void operator()(const N &) const {}
};
// Everything of a color_at node that follows the closest hit `h` of the ray (ro, rd).
template <typename R, uint32_t SF, bool PRIMARY = false, int WALK = WALK_WAVE, bool COLD_LDS = false, typename Children = NoChildrenHook>
RT_DEV Node<R> shade_node(const SceneDev<R> &S, V3<R> ro, V3<R> rd, uint32_t remaining, const Hit<R> &h,
                          Children &&children = Children());

// Scenes with transparency and the sphere hierarchy find the closest hit first and do the container
// bookkeeping afterwards, for the rays that need it (containers_behind); the others do both in one pass.
constexpr bool deferred_containers(uint32_t sf) { return (sf & SF_TRANSPARENT) != 0 && (sf & SF_CULL) != 0; }
// ALWAYS: n1 and n2 of every hit (the probe kernel reports them); otherwise only where they are read:
// the hit object is transparent and the ray may still spawn a refraction ray
template <typename R, uint32_t SF, bool COLD_LDS, bool ALWAYS = false, int WALK = WALK_WAVE>
RT_DEV void containers_if_needed(const SceneDev<R> &S, V3<R> ro, V3<R> rd, uint32_t remaining, Hit<R> &h)
{
    if constexpr (deferred_containers(SF)) {
        bool needed = h.slot != kNone;
        if constexpr (!ALWAYS) needed = needed && remaining != 0u && ColdTable<R, COLD_LDS>::base(S)[h.slot != kNone ? h.obj : 0u].transparency != R(0);
        if (needed) containers_behind<R, (SF & SF_CHAIN) != 0, walk_in_lds(WALK) ? WALK : WALK_WAVE>(S, ro, rd, h);
    }
}
template <typename R, uint32_t SF, bool PRIMARY = false, int WALK = WALK_WAVE, bool COLD_LDS = false, bool ALL_CONTAINERS = false,
          typename Children = NoChildrenHook>
RT_DEV Node<R> trace_node(const SceneDev<R> &S, const PrimScene<R> *P, V3<R> ro, V3<R> rd, uint32_t remaining,
                          Children &&children = Children())
{
    constexpr bool TRACK = (SF & SF_TRANSPARENT) != 0 && !deferred_containers(SF);
    Hit<R> h;
    constexpr bool CULL = (SF & SF_CULL) != 0;
    constexpr bool CHAIN = (SF & SF_CHAIN) != 0;
    if constexpr (PRIMARY) h = closest_hit_primary<R, TRACK, CULL, CHAIN>(S, *P, ro, rd);
    else h = closest_hit<R, TRACK, CULL, CHAIN, WALK, COLD_LDS>(S, ro, rd, Num<R>::inf);
    containers_if_needed<R, SF, COLD_LDS, ALL_CONTAINERS, WALK>(S, ro, rd, remaining, h);
    return shade_node<R, SF, PRIMARY, WALK, COLD_LDS>(S, ro, rd, remaining, h, children);
}

// shade_node in two parts, so that a kernel can answer the shadow queries of a whole wave between
// them: everything up to the shadow ray (which is returned: hit point to light, unit direction,
// distance of the light), and the lighting once it is known whether the point is shadowed.
template <typename R>
struct ShadowQuery {
    V3<R> lightv;
    R ldist;
};
template <typename R, uint32_t SF, bool PRIMARY, bool COLD_LDS, typename Children>
RT_DEV Node<R> shade_prepare(const SceneDev<R> &S, V3<R> ro, V3<R> rd, uint32_t remaining, const Hit<R> &h, ShadowQuery<R> &query,
                             Children &&children)
{
    constexpr bool TRACK = (SF & SF_TRANSPARENT) != 0;
    Node<R> nd;
    nd.has_reflect = nd.has_refract = false;
    nd.lighting = { R(0), R(0), R(0) };
    nd.hit = h.slot != kNone;
    nd.obj = h.obj;
    nd.t = h.t;
    if (!nd.hit) return nd;

    typename ColdTable<R, COLD_LDS>::Rec *const cold_table = ColdTable<R, COLD_LDS>::base(S);
    typename ColdTable<R, COLD_LDS>::Rec &cold = cold_table[h.obj];
    typename ColdTable<R, COLD_LDS>::Index *const slot_obj = ColdTable<R, COLD_LDS>::slot_obj(S);

    // prepare_computations, world.rs:242-251; Ray::position, rays.rs:24-28
    V3<R> pt = { ro.x + rd.x * h.t, ro.y + rd.y * h.t, ro.z + rd.z * h.t };
    V3<R> eyev = { -rd.x, -rd.y, -rd.z };
    // Object::normal_at, objects.rs:172-183
    V3<R> lp = xform_point(cold.inv, pt);
    V3<R> ln = local_normal(cold.type, cold.ymin, cold.ymax, lp);
    V3<R> wn = { (cold.inv_t[0] * ln.x + cold.inv_t[1] * ln.y) + cold.inv_t[2] * ln.z,
                 (cold.inv_t[3] * ln.x + cold.inv_t[4] * ln.y) + cold.inv_t[5] * ln.z,
                 (cold.inv_t[6] * ln.x + cold.inv_t[7] * ln.y) + cold.inv_t[8] * ln.z };
    V3<R> normalv = normalize3(wn);
    bool inside = false;
    if (dot3(normalv, eyev) < R(0)) {
        inside = true;
        normalv = { -normalv.x, -normalv.y, -normalv.z };
    }
    // Tuple::reflect_vector, tuples.rs:87-91
    R d2 = R(2) * dot3(rd, normalv);
    V3<R> reflectv = { rd.x - normalv.x * d2, rd.y - normalv.y * d2, rd.z - normalv.z * d2 };
    // Computations::new, world.rs:219-220
    R off = Num<R>::eps;
    if constexpr (!Num<R>::exact) {
        // The reference's 1e-5 is kept wherever floats resolve it (|p| <= 4: at least 20 ulp);
        // beyond that it grows in proportion to |p|, since 1e-5 falls below the error of an f32
        // hit point and, past |p| ~ 100, below the spacing of floats.  Inflating it near the
        // origin as well is visible: a ray refracted into a sphere at grazing incidence starts
        // deeper and lands 30-100 offsets away from where the reference's ray does.
        const R mag = Num<R>::max_(Num<R>::max_(Num<R>::abs_(pt.x), Num<R>::abs_(pt.y)), Num<R>::abs_(pt.z));
        off = Num<R>::eps * Num<R>::max_(R(1), mag * R(0.25));
    }
    V3<R> over = { pt.x + normalv.x * off, pt.y + normalv.y * off, pt.z + normalv.z * off };
    V3<R> under = { pt.x - normalv.x * off, pt.y - normalv.y * off, pt.z - normalv.z * off };

    R n1 = R(1), n2 = R(1);
    if constexpr (TRACK) {
        if (h.top1 != kNone) n1 = cold_table[slot_obj[h.top1]].refractive_index;
        if (!h.hit_open) n2 = cold.refractive_index;
        else if (h.top1 != h.slot) n2 = cold_table[slot_obj[h.top1]].refractive_index;
        else if (h.top2 != kNone) n2 = cold_table[slot_obj[h.top2]].refractive_index;
    }

    nd.point = pt; nd.normalv = normalv; nd.over_point = over; nd.under_point = under; nd.reflectv = reflectv;
    nd.inside = inside; nd.n1 = n1; nd.n2 = n2;

    // reflected_color, world.rs:160-173
    [[maybe_unused]] R reflective = R(0), transparency = R(0);
    if constexpr ((SF & SF_REFLECTIVE) != 0) {
        reflective = cold.reflective;
        if (remaining >= 1 && reflective != R(0)) {
            nd.has_reflect = true;
            nd.reflect_factor = reflective;
        }
    }
    // refracted_color, world.rs:175-190
    if constexpr ((SF & SF_TRANSPARENT) != 0) {
        transparency = cold.transparency;
        if (transparency != R(0) && remaining != 0) {
            R n_ratio = Num<R>::div(n1, n2);
            R cos_i = dot3(eyev, normalv);
            R sin2_t = (n_ratio * n_ratio) * (R(1) - (cos_i * cos_i));
            if (!(sin2_t > R(1))) {
                R cos_t = Num<R>::sqrt_(R(1) - sin2_t);
                R k = (n_ratio * cos_i) - cos_t;
                nd.refract_dir = { normalv.x * k - eyev.x * n_ratio, normalv.y * k - eyev.y * n_ratio,
                                   normalv.z * k - eyev.z * n_ratio };
                nd.has_refract = true;
                nd.refract_factor = transparency;
            }
        }
    }
    // Schlick mix, world.rs:128-131 and 227-239
    if constexpr ((SF & SF_REFLECTIVE) != 0 && (SF & SF_TRANSPARENT) != 0) {
        if (reflective > R(0) && transparency > R(0)) {
            R cosv = dot3(eyev, normalv);
            R reflectance;
            bool tir = false;
            if (n1 > n2) {
                R n = Num<R>::div(n1, n2);
                R sin2_t = (n * n) * (R(1) - (cosv * cosv));
                if (sin2_t > R(1)) tir = true;
                else cosv = Num<R>::sqrt_(R(1) - sin2_t);
            }
            if (tir) {
                reflectance = R(1);
            } else {
                R q = Num<R>::div(n1 - n2, n1 + n2);
                R r0 = q * q;
                R x = R(1) - cosv;
                R x2 = x * x;
                R x4 = x2 * x2;
                reflectance = r0 + (R(1) - r0) * (x * x4);
            }
            nd.reflect_factor = reflective * reflectance;
            nd.refract_factor = transparency * (R(1) - reflectance);
        }
    }
    children(nd);

    // shade_hit, world.rs:121-123.  is_shadowed (world.rs:146-150) and lighting (lights.rs:49)
    // both normalise light.position - over_point; the vector is formed once here.
    V3<R> lvec = { S.light_pos[0] - over.x, S.light_pos[1] - over.y, S.light_pos[2] - over.z };
    query.lightv = normalized(lvec, (lvec.x * lvec.x + lvec.y * lvec.y) + lvec.z * lvec.z, &query.ldist);
    return nd;
}

template <typename R, uint32_t SF, bool PRIMARY, bool COLD_LDS>
RT_DEV void shade_finish(const SceneDev<R> &S, V3<R> rd, Node<R> &nd, const ShadowQuery<R> &query, bool shadowed)
{
    typename ColdTable<R, COLD_LDS>::Rec &cold = ColdTable<R, COLD_LDS>::base(S)[nd.obj];
    const V3<R> eyev = { -rd.x, -rd.y, -rd.z };
    const V3<R> over = nd.over_point, normalv = nd.normalv, lightv = query.lightv;
    nd.shadowed = shadowed;
    // lighting, lights.rs:43-78, evaluated at over_point
    V3<R> clr = { cold.color[0], cold.color[1], cold.color[2] };
    if constexpr ((SF & SF_PATTERN) != 0) {
        if (cold.pattern_type != PT_SOLID) {  // Object::pattern_at_object, objects.rs:200-207
            V3<R> op = xform_point(cold.inv, over);
            V3<R> pp = xform_point(cold.pinv, op);
            clr = pattern_at<R>(cold.pattern_type, { cold.c1[0], cold.c1[1], cold.c1[2] },
                                { cold.c2[0], cold.c2[1], cold.c2[2] }, pp);
        }
    }
    V3<R> li = { S.light_int[0], S.light_int[1], S.light_int[2] };
    V3<R> eff = { clr.x * li.x, clr.y * li.y, clr.z * li.z };
    V3<R> result = { eff.x * cold.ambient, eff.y * cold.ambient, eff.z * cold.ambient };
    if (!shadowed) {
        V3<R> diffuse = { R(0), R(0), R(0) }, specular = { R(0), R(0), R(0) };
        R ldn = dot3(lightv, normalv);
        if (ldn >= R(0)) {
            diffuse = { (eff.x * cold.diffuse) * ldn, (eff.y * cold.diffuse) * ldn, (eff.z * cold.diffuse) * ldn };
            R l2 = R(2) * dot3(lightv, normalv);
            V3<R> rv = { -(lightv.x - normalv.x * l2), -(lightv.y - normalv.y * l2), -(lightv.z - normalv.z * l2) };
            R rde = dot3(rv, eyev);
            if (!(rde <= R(0))) {
                R factor = Num<R>::pow_(rde, cold.shininess);
                specular = { (li.x * cold.specular) * factor, (li.y * cold.specular) * factor,
                             (li.z * cold.specular) * factor };
            }
        }
        result = { (result.x + diffuse.x) + specular.x, (result.y + diffuse.y) + specular.y,
                   (result.z + diffuse.z) + specular.z };
    }
    nd.lighting = result;
}

template <typename R, uint32_t SF, bool PRIMARY, int WALK, bool COLD_LDS, typename Children>
RT_DEV Node<R> shade_node(const SceneDev<R> &S, V3<R> ro, V3<R> rd, uint32_t remaining, const Hit<R> &h, Children &&children)
{
    ShadowQuery<R> query;
    Node<R> nd = shade_prepare<R, SF, PRIMARY, COLD_LDS>(S, ro, rd, remaining, h, query, children);
    if (!nd.hit) return nd;
    const bool shadowed = is_shadowed<R, SF, WALK, COLD_LDS>(S, nd.over_point, query.lightv, query.ldist);
    shade_finish<R, SF, PRIMARY, COLD_LDS>(S, rd, nd, query, shadowed);
    return nd;
}



// src/scenes/camera.rs:147-158
template <typename R>
RT_DEV void ray_for_pixel(const CameraDev<R> &cam, uint32_t x, uint32_t y, V3<R> &ro, V3<R> &rd)
{
    R xoffset = (R(x) + R(0.5)) * cam.pixel_size;
    R yoffset = (R(y) + R(0.5)) * cam.pixel_size;
    R world_x = cam.half_width - xoffset;
    R world_y = cam.half_height - yoffset;
    // inverse_transform * point(world_x, world_y, -1): m[i][2] * -1.0 is exactly -m[i][2]
    V3<R> pixel = { ((cam.inv[0] * world_x + cam.inv[1] * world_y) - cam.inv[2]) + cam.inv[3],
                    ((cam.inv[4] * world_x + cam.inv[5] * world_y) - cam.inv[6]) + cam.inv[7],
                    ((cam.inv[8] * world_x + cam.inv[9] * world_y) - cam.inv[10]) + cam.inv[11] };
    ro = { cam.origin[0], cam.origin[1], cam.origin[2] };
    rd = normalize3<R>({ pixel.x - ro.x, pixel.y - ro.y, pixel.z - ro.z });
}

}  // namespace rtx
