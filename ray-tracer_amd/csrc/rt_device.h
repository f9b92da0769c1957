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
    static constexpr double eps = 0.00001;                   // src/utils.rs:1
    static constexpr double big = 1.7976931348623157e308;    // f64::MAX
    static constexpr double inf = __builtin_huge_val();
    static RT_DEV double sqrt_(double x) { return __builtin_sqrt(x); }
    static RT_DEV double div(double a, double b) { return a / b; }
    static RT_DEV double pow_(double a, double b) { return pow(a, b); }
    static RT_DEV double abs_(double x) { return __builtin_fabs(x); }
    static RT_DEV double floor_(double x) { return __builtin_floor(x); }
    static RT_DEV double max_(double a, double b) { return __builtin_fmax(a, b); }
    static RT_DEV double min_(double a, double b) { return __builtin_fmin(a, b); }
};
template <> struct Num<float> {
    static constexpr float eps = 0.00001f;
    static constexpr float big = 3.402823466e38f;
    static constexpr float inf = __builtin_huge_valf();
    static RT_DEV float sqrt_(float x) { return __builtin_sqrtf(x); }
    static RT_DEV float div(float a, float b) { return a / b; }
    static RT_DEV float pow_(float a, float b) { return powf(a, b); }
    static RT_DEV float abs_(float x) { return __builtin_fabsf(x); }
    static RT_DEV float floor_(float x) { return __builtin_floorf(x); }
    static RT_DEV float max_(float a, float b) { return __builtin_fmaxf(a, b); }
    static RT_DEV float min_(float a, float b) { return __builtin_fminf(a, b); }
};

template <typename R> struct V3 { R x, y, z; };

// src/matrices/tuples.rs:145-154
template <typename R> RT_DEV R dot3(V3<R> a, V3<R> b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// src/matrices/tuples.rs:58-73: magnitude, then three divisions
template <typename R> RT_DEV V3<R> normalize3(V3<R> a)
{
    R mag = Num<R>::sqrt_((a.x * a.x + a.y * a.y) + a.z * a.z);
    return { Num<R>::div(a.x, mag), Num<R>::div(a.y, mag), Num<R>::div(a.z, mag) };
}
// src/matrices/matrix4.rs:172-193 for a point (w = 1) and a vector (w = 0); m = 3 rows x 4
template <typename R> RT_DEV V3<R> xform_point(const R *m, V3<R> p)
{
    return { ((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3],
             ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
             ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11] };
}
template <typename R> RT_DEV V3<R> xform_vector(const R *m, V3<R> v)
{
    return { (m[0] * v.x + m[1] * v.y) + m[2] * v.z,
             (m[4] * v.x + m[5] * v.y) + m[6] * v.z,
             (m[8] * v.x + m[9] * v.y) + m[10] * v.z };
}

// ---- ray/object intersection in object space -----------------------------------------
// Each routine calls emit(t) once per intersection, in the order the reference pushes them.

// src/shapes/spheres.rs:24-42
template <typename R, typename Emit> RT_DEV void isect_sphere(V3<R> o, V3<R> d, Emit &&emit)
{
    R a = dot3(d, d);
    R b = R(2) * dot3(d, o);
    R c = dot3(o, o) - R(1);
    R disc = b * b - ((R(4) * a) * c);
    if (!(disc < R(0))) {
        R sq = Num<R>::sqrt_(disc);
        R two_a = R(2) * a;
        R t1 = Num<R>::div(-b - sq, two_a);
        R t2 = Num<R>::div(-b + sq, two_a);
        emit(t1);
        emit(t2);
    }
}
// src/shapes/planes.rs:11-17
template <typename R, typename Emit> RT_DEV void isect_plane(V3<R> o, V3<R> d, Emit &&emit)
{
    if (!(Num<R>::abs_(d.y) < Num<R>::eps)) emit(Num<R>::div(-o.y, d.y));
}
// src/shapes/cubes.rs:31-51
template <typename R> RT_DEV void cube_axis(R origin, R direction, R &tmin, R &tmax)
{
    R tmin_num = R(-1) - origin;
    R tmax_num = R(1) - origin;
    R lo, hi;
    if (!(Num<R>::abs_(direction) < Num<R>::eps)) {
        lo = Num<R>::div(tmin_num, direction);
        hi = Num<R>::div(tmax_num, direction);
    } else {
        lo = tmin_num * Num<R>::big;
        hi = tmax_num * Num<R>::big;
    }
    if (lo > hi) { R tmp = lo; lo = hi; hi = tmp; }
    tmin = lo;
    tmax = hi;
}
// src/shapes/cubes.rs:7-29
template <typename R, typename Emit> RT_DEV void isect_cube(V3<R> o, V3<R> d, Emit &&emit)
{
    R xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
    cube_axis(o.x, d.x, xtmin, xtmax);
    cube_axis(o.y, d.y, ytmin, ytmax);
    if (ytmin > xtmax || xtmin > ytmax) return;
    cube_axis(o.z, d.z, ztmin, ztmax);
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
        if (x * x + z * z <= lim) emit(t1);
    }
    R t2 = Num<R>::div(ymax - o.y, d.y);
    {
        R x = o.x + (t2 * d.x);
        R z = o.z + (t2 * d.z);
        R lim = CONE ? Num<R>::abs_(ymax) : R(1);
        if (x * x + z * z <= lim) emit(t2);
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
    if (ymin < y0 && y0 < ymax) emit(t0);
    R y1 = o.y + (t1 * d.y);
    if (ymin < y1 && y1 < ymax) emit(t1);
    return true;
}
// src/shapes/cylinders.rs:27-64
template <typename R, typename Emit>
RT_DEV void isect_cylinder(V3<R> o, V3<R> d, R ymin, R ymax, bool closed, Emit &&emit)
{
    R a = d.x * d.x + d.z * d.z;
    if (a < Num<R>::eps) {
        isect_caps<R, false>(o, d, ymin, ymax, closed, emit);
        return;
    }
    R b = (R(2) * o.x) * d.x + (R(2) * o.z) * d.z;
    R c = (o.x * o.x + o.z * o.z) - R(1);
    if (!isect_wall(a, b, c, o, d, ymin, ymax, emit)) return;
    isect_caps<R, false>(o, d, ymin, ymax, closed, emit);
}
// src/shapes/cones.rs:27-82
template <typename R, typename Emit>
RT_DEV void isect_cone(V3<R> o, V3<R> d, R ymin, R ymax, bool closed, Emit &&emit)
{
    R a = (d.x * d.x - d.y * d.y) + d.z * d.z;
    R b = ((R(2) * o.x) * d.x - (R(2) * o.y) * d.y) + (R(2) * o.z) * d.z;
    bool a0 = Num<R>::abs_(a) < Num<R>::eps;
    bool b0 = Num<R>::abs_(b) < Num<R>::eps;
    if (a0 && b0) {
        isect_caps<R, true>(o, d, ymin, ymax, closed, emit);
        return;
    }
    R c = (o.x * o.x - o.y * o.y) + o.z * o.z;
    if (a0) {
        emit(Num<R>::div(-c, R(2) * b));
        isect_caps<R, true>(o, d, ymin, ymax, closed, emit);
        return;
    }
    if (!isect_wall(a, b, c, o, d, ymin, ymax, emit)) return;
    isect_caps<R, true>(o, d, ymin, ymax, closed, emit);
}

// src/shapes/objects.rs:185-198 with src/rays.rs:30-34: ray into object space (direction not
// renormalised), then the per-shape routine.  `rec` is addressed with a wave-uniform index.
template <typename R, typename Emit>
RT_DEV void isect_object(const HotRec<R> &rec, V3<R> ro, V3<R> rd, Emit &&emit)
{
    const uint32_t type = rec.type;
    if (type == SH_PLANE) {
        // only the y row of the object-space ray is read by planes.rs:11-17
        V3<R> o = { R(0), ((rec.inv[4] * ro.x + rec.inv[5] * ro.y) + rec.inv[6] * ro.z) + rec.inv[7], R(0) };
        V3<R> d = { R(0), (rec.inv[4] * rd.x + rec.inv[5] * rd.y) + rec.inv[6] * rd.z, R(0) };
        isect_plane(o, d, emit);
        return;
    }
    V3<R> o = xform_point(rec.inv, ro);
    V3<R> d = xform_vector(rec.inv, rd);
    if (type == SH_SPHERE) isect_sphere(o, d, emit);
    else if (type == SH_CUBE) isect_cube(o, d, emit);
    else if (type == SH_CYLINDER) isect_cylinder(o, d, rec.ymin, rec.ymax, (rec.flags & OF_CLOSED) != 0, emit);
    else isect_cone(o, d, rec.ymin, rec.ymax, (rec.flags & OF_CLOSED) != 0, emit);
}

// ---- closest hit over the whole world ---------------------------------------------------
// The reference concatenates every object's intersections, stable-sorts by t and takes the
// first t >= 0 (src/scenes/world.rs:98-119).  Equivalent single pass: the smallest t >= 0,
// the lowest object index winning ties (strict '<' while walking objects in order).
//
// With TRACK the same pass also derives n1/n2 of prepare_computations
// (world.rs:253-285).  Everything that precedes the hit in the sorted list has t < 0.  An
// object sits in `containers` at the hit iff it has an odd number of negative t's, and the
// list is ordered by each object's largest negative t (ties: object order).  Only the last
// entry and, when the hit object itself is that entry, the one before it are ever read, so
// the two "latest" open objects are all that must be remembered.
template <typename R>
struct Hit {
    R t;
    uint32_t obj;      // 0xffffffff = miss
    bool hit_open;     // TRACK: the hit object is in `containers` before it is toggled
    R top1_t, top2_t;  // TRACK: largest negative t of the last / second-to-last container
    uint32_t top1, top2;
};

template <typename R, bool TRACK>
RT_DEV Hit<R> closest_hit(const SceneDev<R> &S, V3<R> ro, V3<R> rd)
{
    Hit<R> h;
    h.t = Num<R>::inf;
    h.obj = 0xffffffffu;
    h.hit_open = false;
    h.top1 = h.top2 = 0xffffffffu;
    h.top1_t = h.top2_t = -Num<R>::inf;
    for (uint32_t i = 0; i < S.n; ++i) {
        const HotRec<R> &rec = S.hot[i];
        if constexpr (!TRACK) {
            isect_object(rec, ro, rd, [&](R t) {
                if (t >= R(0) && t < h.t) { h.t = t; h.obj = i; }
            });
        } else {
            R best = Num<R>::inf;       // smallest t >= 0 of this object
            R last_neg = -Num<R>::inf;  // largest t < 0 of this object
            uint32_t n_neg = 0;
            isect_object(rec, ro, rd, [&](R t) {
                if (t >= R(0)) best = Num<R>::min_(best, t);
                else if (t < R(0)) { n_neg++; last_neg = Num<R>::max_(last_neg, t); }
            });
            const bool open = (n_neg & 1u) && (rec.flags & OF_CONTAINER);
            if (best < h.t) { h.t = best; h.obj = i; h.hit_open = open; }
            if (open) {
                // later in the sorted list = larger t, or equal t and higher object index
                if (last_neg >= h.top1_t) {
                    h.top2 = h.top1; h.top2_t = h.top1_t;
                    h.top1 = i; h.top1_t = last_neg;
                } else if (last_neg >= h.top2_t) {
                    h.top2 = i; h.top2_t = last_neg;
                }
            }
        }
    }
    return h;
}

// src/scenes/world.rs:146-158.  shadowed iff the closest hit (t >= 0) lies before the light
// and its object casts a shadow.  When every object casts, any hit with 0 <= t < distance
// decides it, so the walk needs no closest-hit bookkeeping.
template <typename R, uint32_t SF>
RT_DEV bool is_shadowed(const SceneDev<R> &S, V3<R> p)
{
    V3<R> v = { S.light_pos[0] - p.x, S.light_pos[1] - p.y, S.light_pos[2] - p.z };
    R distance = Num<R>::sqrt_((v.x * v.x + v.y * v.y) + v.z * v.z);
    V3<R> dir = { Num<R>::div(v.x, distance), Num<R>::div(v.y, distance), Num<R>::div(v.z, distance) };
    if constexpr ((SF & SF_ALL_CAST) != 0) {
        bool shadowed = false;
        for (uint32_t i = 0; i < S.n; ++i) {
            isect_object(S.hot[i], p, dir, [&](R t) {
                if (t >= R(0) && t < distance) shadowed = true;
            });
        }
        return shadowed;
    } else {
        Hit<R> h = closest_hit<R, false>(S, p, dir);
        if (h.obj == 0xffffffffu) return false;
        return h.t < distance && (S.hot[h.obj].flags & OF_CASTS_SHADOW) != 0;
    }
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

template <typename R, uint32_t SF>
RT_DEV Node<R> trace_node(const SceneDev<R> &S, V3<R> ro, V3<R> rd, uint32_t remaining)
{
    constexpr bool TRACK = (SF & SF_TRANSPARENT) != 0;
    Node<R> nd;
    nd.has_reflect = nd.has_refract = false;
    nd.lighting = { R(0), R(0), R(0) };
    Hit<R> h = closest_hit<R, TRACK>(S, ro, rd);
    nd.hit = h.obj != 0xffffffffu;
    nd.obj = h.obj;
    nd.t = h.t;
    if (!nd.hit) return nd;

    const HotRec<R> &hot = S.hot[h.obj];
    const ColdRec<R> &cold = S.cold[h.obj];

    // prepare_computations, world.rs:242-251; Ray::position, rays.rs:24-28
    V3<R> pt = { ro.x + rd.x * h.t, ro.y + rd.y * h.t, ro.z + rd.z * h.t };
    V3<R> eyev = { -rd.x, -rd.y, -rd.z };
    // Object::normal_at, objects.rs:172-183
    V3<R> lp = xform_point(hot.inv, pt);
    V3<R> ln = local_normal(hot.type, hot.ymin, hot.ymax, lp);
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
    V3<R> over = { pt.x + normalv.x * Num<R>::eps, pt.y + normalv.y * Num<R>::eps, pt.z + normalv.z * Num<R>::eps };
    V3<R> under = { pt.x - normalv.x * Num<R>::eps, pt.y - normalv.y * Num<R>::eps, pt.z - normalv.z * Num<R>::eps };

    R n1 = R(1), n2 = R(1);
    if constexpr (TRACK) {
        if (h.top1 != 0xffffffffu) n1 = S.cold[h.top1].refractive_index;
        if (!h.hit_open) n2 = cold.refractive_index;
        else if (h.top1 != h.obj) n2 = S.cold[h.top1].refractive_index;
        else if (h.top2 != 0xffffffffu) n2 = S.cold[h.top2].refractive_index;
    }

    nd.point = pt; nd.normalv = normalv; nd.over_point = over; nd.under_point = under; nd.reflectv = reflectv;
    nd.inside = inside; nd.n1 = n1; nd.n2 = n2;

    // shade_hit, world.rs:121-123
    bool shadowed = is_shadowed<R, SF>(S, over);
    nd.shadowed = shadowed;

    // lighting, lights.rs:43-78, evaluated at over_point
    V3<R> clr = { cold.color[0], cold.color[1], cold.color[2] };
    if constexpr ((SF & SF_PATTERN) != 0) {
        if (cold.pattern_type != PT_SOLID) {  // Object::pattern_at_object, objects.rs:200-207
            V3<R> op = xform_point(hot.inv, over);
            V3<R> pp = xform_point(cold.pinv, op);
            clr = pattern_at<R>(cold.pattern_type, { cold.c1[0], cold.c1[1], cold.c1[2] },
                                { cold.c2[0], cold.c2[1], cold.c2[2] }, pp);
        }
    }
    V3<R> li = { S.light_int[0], S.light_int[1], S.light_int[2] };
    V3<R> eff = { clr.x * li.x, clr.y * li.y, clr.z * li.z };
    V3<R> lightv = normalize3<R>({ S.light_pos[0] - over.x, S.light_pos[1] - over.y, S.light_pos[2] - over.z });
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

    // reflected_color, world.rs:160-173
    R reflective = R(0), transparency = R(0);
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
