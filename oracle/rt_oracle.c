/*
 * rt_oracle.c -- CPU restatement of the per-pixel render loop of malloc-man/ray-tracer.
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Recursive, list-based and sorted exactly
 * like the reference so that it can be checked by reading it next to the Rust source.
 * Parity: pinned to the reference's unit-test constants (1e-5); agreement with the Rust
 * binary beyond that is unpinned because the reference cannot be built here.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread rt_oracle.c -lm
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* utils.rs:1 */
#define EPSILON 0.00001
/* world.rs:3 */
#define DEFAULT_RECURSION_DEPTH 5

/* ------------------------------------------------------------------------------------
 * panics: the reference aborts the render thread with panic!/unwrap; here a panic
 * unwinds to the API boundary and the call returns ORC_PANIC.
 * ---------------------------------------------------------------------------------- */
static __thread jmp_buf *tl_jmp = NULL;
static __thread char tl_panic[256];

const char *orc_last_panic(void) { return tl_panic; }

static void panic(const char *msg)
{
    snprintf(tl_panic, sizeof tl_panic, "%s", msg);
    if (tl_jmp) longjmp(*tl_jmp, 1);
    fprintf(stderr, "rt_oracle: panic outside an API boundary: %s\n", msg);
    abort();
}

#define API_ENTER()                                 \
    jmp_buf jb_;                                    \
    jmp_buf *saved_jb_ = tl_jmp;                    \
    if (setjmp(jb_)) {                              \
        tl_jmp = saved_jb_;                         \
        return ORC_PANIC;                           \
    }                                               \
    tl_jmp = &jb_
#define API_LEAVE() tl_jmp = saved_jb_

/* utils.rs:7-11 */
static int approx_eq(double a, double b) { return fabs(a - b) < EPSILON; }

/* ------------------------------------------------------------------------------------
 * tuples (matrices/tuples.rs)
 * ---------------------------------------------------------------------------------- */
typedef struct { double x, y, z; int v; } tuple_t;

/* tuples.rs:12-20 */
static tuple_t mk_tuple(double x, double y, double z, int v)
{
    if (v != 0 && v != 1) panic("Attempted to create tuple with v not 0 or 1");
    tuple_t t = { x, y, z, v };
    return t;
}
static tuple_t point(double x, double y, double z) { tuple_t t = { x, y, z, 1 }; return t; }
static tuple_t vector(double x, double y, double z) { tuple_t t = { x, y, z, 0 }; return t; }

/* tuples.rs:58-64: x^2 + y^2 + z^2 + v^2 with v == 0 */
static double magnitude(tuple_t a)
{
    if (a.v != 0) panic("Cannot get magnitude of point");
    return sqrt(a.x * a.x + a.y * a.y + a.z * a.z + (double)(a.v * a.v));
}
/* tuples.rs:66-73: three divisions */
static tuple_t normalize(tuple_t a)
{
    if (a.v != 0) panic("Cannot normalize a point");
    double mag = magnitude(a);
    return vector(a.x / mag, a.y / mag, a.z / mag);
}
/* tuples.rs:76-85 */
static tuple_t xprod(tuple_t a, tuple_t b)
{
    if (a.v != 0 || b.v != 0) panic("Need two vectors to compute cross product");
    return vector(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* tuples.rs:106-118 */
static tuple_t t_add(tuple_t a, tuple_t b) { return mk_tuple(a.x + b.x, a.y + b.y, a.z + b.z, a.v + b.v); }
static tuple_t t_sub(tuple_t a, tuple_t b) { return mk_tuple(a.x - b.x, a.y - b.y, a.z - b.z, a.v - b.v); }
/* tuples.rs:121-130 */
static tuple_t t_neg(tuple_t a)
{
    if (a.v != 0) panic("Cannot negate a point");
    return mk_tuple(-a.x, -a.y, -a.z, -a.v);
}
/* tuples.rs:133-142 */
static tuple_t t_scale(tuple_t a, double s)
{
    if (a.v != 0) panic("Cannot multiply a point");
    return vector(a.x * s, a.y * s, a.z * s);
}
/* tuples.rs:145-154 */
static double dot(tuple_t a, tuple_t b)
{
    if (a.v != 0 || b.v != 0) panic("Need two vectors to compute dot product");
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
/* tuples.rs:87-91 */
static tuple_t reflect_vector(tuple_t v, tuple_t normal)
{
    double d = 2.0 * dot(v, normal);
    tuple_t nrm = t_scale(normal, d);
    return t_sub(v, nrm);
}

/* ------------------------------------------------------------------------------------
 * colours (surfaces/colors.rs:49-81)
 * ---------------------------------------------------------------------------------- */
typedef struct { double r, g, b; } color_t;
static color_t mk_color(double r, double g, double b) { color_t c = { r, g, b }; return c; }
static color_t c_add(color_t a, color_t b) { return mk_color(a.r + b.r, a.g + b.g, a.b + b.b); }
static color_t c_sub(color_t a, color_t b) { return mk_color(a.r - b.r, a.g - b.g, a.b - b.b); }
static color_t c_mul(color_t a, color_t b) { return mk_color(a.r * b.r, a.g * b.g, a.b * b.b); }
static color_t c_scale(color_t a, double s) { return mk_color(a.r * s, a.g * s, a.b * s); }
/* colors.rs:83-92 */
static int color_eq(color_t a, color_t b) { return approx_eq(a.r, b.r) && approx_eq(a.g, b.g) && approx_eq(a.b, b.b); }

/* ------------------------------------------------------------------------------------
 * matrices (matrices/matrix2.rs, matrix3.rs, matrix4.rs)
 * ---------------------------------------------------------------------------------- */
typedef struct { double m[4][4]; } mat4_t;

static mat4_t m4_zero(void) { mat4_t r; memset(&r, 0, sizeof r); return r; }
/* matrix4.rs:22-28 */
static mat4_t m4_identity(void)
{
    mat4_t r = m4_zero();
    for (int i = 0; i < 4; i++) r.m[i][i] = 1.0;
    return r;
}
/* matrix4.rs:44-52 */
static mat4_t m4_transpose(mat4_t a)
{
    mat4_t r = m4_zero();
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[j][i] = a.m[i][j];
    return r;
}
/* matrix2.rs:23-25 */
static double det2(double m[2][2]) { return m[0][0] * m[1][1] - m[1][0] * m[0][1]; }
/* matrix3.rs:23-45 */
static void sub3(double m[3][3], int row, int col, double out[2][2])
{
    int ci = 0;
    for (int i = 0; i < 3; i++) {
        if (i == row) continue;
        int cj = 0;
        for (int j = 0; j < 3; j++) {
            if (j == col) continue;
            out[ci][cj] = m[i][j];
            cj++;
        }
        ci++;
    }
}
/* matrix3.rs:47-59 */
static double cofactor3(double m[3][3], int row, int col)
{
    double s[2][2];
    sub3(m, row, col, s);
    double minor = det2(s);
    return ((row + col) % 2 == 1) ? -minor : minor;
}
/* matrix3.rs:61-67: expansion along the first column, accumulated from 0.0 */
static double det3(double m[3][3])
{
    double d = 0.0;
    for (int i = 0; i < 3; i++) d += m[i][0] * cofactor3(m, i, 0);
    return d;
}
/* matrix4.rs:54-76 */
static void sub4(const mat4_t *a, int row, int col, double out[3][3])
{
    int ci = 0;
    for (int i = 0; i < 4; i++) {
        if (i == row) continue;
        int cj = 0;
        for (int j = 0; j < 4; j++) {
            if (j == col) continue;
            out[ci][cj] = a->m[i][j];
            cj++;
        }
        ci++;
    }
}
/* matrix4.rs:78-90 */
static double cofactor4(const mat4_t *a, int row, int col)
{
    double s[3][3];
    sub4(a, row, col, s);
    double minor = det3(s);
    return ((row + col) % 2 == 1) ? -minor : minor;
}
/* matrix4.rs:92-98 */
static double det4(const mat4_t *a)
{
    double d = 0.0;
    for (int i = 0; i < 4; i++) d += a->m[i][0] * cofactor4(a, i, 0);
    return d;
}
/* matrix4.rs:100-119 */
static mat4_t m4_invert(mat4_t a)
{
    double d = det4(&a);
    if (approx_eq(d, 0.0)) panic("Attempted to invert a non-invertible matrix");
    double determinant = det4(&a);
    mat4_t r = m4_zero();
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double c = cofactor4(&a, i, j);
            r.m[j][i] = c / determinant;
        }
    return r;
}
/* matrix4.rs:155-170 */
static mat4_t m4_mul(mat4_t a, mat4_t b)
{
    mat4_t r = m4_zero();
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double val = 0.0;
            for (int k = 0; k < 4; k++) val += a.m[i][k] * b.m[k][j];
            r.m[i][j] = val;
        }
    return r;
}
/* matrix4.rs:172-193: all four rows are accumulated from 0.0, the fourth is discarded */
static tuple_t m4_mul_tuple(const mat4_t *a, tuple_t t)
{
    double other[4] = { t.x, t.y, t.z, (double)t.v };
    double col[4];
    for (int i = 0; i < 4; i++) {
        double val = 0.0;
        for (int k = 0; k < 4; k++) val += a->m[i][k] * other[k];
        col[i] = val;
    }
    tuple_t r = { col[0], col[1], col[2], t.v };
    return r;
}
/* matrix4.rs:142-153 */
static int m4_eq(const mat4_t *a, const mat4_t *b)
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            if (!approx_eq(a->m[i][j], b->m[i][j])) return 0;
    return 1;
}

/* transformations.rs:3-61 */
static mat4_t translation(double x, double y, double z)
{
    mat4_t r = m4_identity();
    r.m[0][3] = x; r.m[1][3] = y; r.m[2][3] = z;
    return r;
}
static mat4_t scaling(double x, double y, double z)
{
    mat4_t r = m4_zero();
    r.m[0][0] = x; r.m[1][1] = y; r.m[2][2] = z; r.m[3][3] = 1.0;
    return r;
}
static mat4_t rotation_x(double rad)
{
    mat4_t r = m4_identity();
    r.m[1][1] = cos(rad); r.m[1][2] = -(sin(rad));
    r.m[2][1] = sin(rad); r.m[2][2] = cos(rad);
    return r;
}
static mat4_t rotation_y(double rad)
{
    mat4_t r = m4_identity();
    r.m[0][0] = cos(rad); r.m[0][2] = sin(rad);
    r.m[2][0] = -(sin(rad)); r.m[2][2] = cos(rad);
    return r;
}
static mat4_t rotation_z(double rad)
{
    mat4_t r = m4_identity();
    r.m[0][0] = cos(rad); r.m[0][1] = -(sin(rad));
    r.m[1][0] = sin(rad); r.m[1][1] = cos(rad);
    return r;
}
static mat4_t shearing(double xy, double xz, double yx, double yz, double zx, double zy)
{
    mat4_t r = m4_identity();
    r.m[0][1] = xy; r.m[0][2] = xz;
    r.m[1][0] = yx; r.m[1][2] = yz;
    r.m[2][0] = zx; r.m[2][1] = zy;
    return r;
}
/* transformations.rs:63-77 */
static mat4_t view_transform(tuple_t from, tuple_t to, tuple_t up)
{
    tuple_t forward = normalize(t_sub(to, from));
    tuple_t left = xprod(forward, normalize(up));
    tuple_t true_up = xprod(left, forward);
    mat4_t o = m4_identity();
    o.m[0][0] = left.x; o.m[0][1] = left.y; o.m[0][2] = left.z;
    o.m[1][0] = true_up.x; o.m[1][1] = true_up.y; o.m[1][2] = true_up.z;
    o.m[2][0] = -forward.x; o.m[2][1] = -forward.y; o.m[2][2] = -forward.z;
    return m4_mul(o, translation(-from.x, -from.y, -from.z));
}

/* ------------------------------------------------------------------------------------
 * patterns, materials, objects
 * ---------------------------------------------------------------------------------- */
/* patterns.rs:4-10 */
typedef struct {
    int type;
    mat4_t transform, inverse;
    color_t c1, c2;
} pattern_t;
/* materials.rs:4-15 */
typedef struct {
    color_t color;
    double ambient, diffuse, specular, shininess;
    pattern_t pattern;
    double reflective, transparency, refractive_index;
    int casts_shadow;
} material_t;
/* objects.rs:4-20 */
struct orc_object {
    material_t material;
    mat4_t transform, inverse, inverse_t;
    int shape;
    double ymin, ymax; /* Cylinder / Cone payload */
    int closed;
    double tlist[15];
};

/* patterns.rs:36-44 */
static pattern_t pattern_new(int type, color_t c1, color_t c2)
{
    pattern_t p;
    p.type = type;
    p.transform = m4_identity();
    p.inverse = m4_identity();
    p.c1 = c1;
    p.c2 = c2;
    return p;
}
/* materials.rs:18-31; patterns.rs:120-122 */
static material_t material_new(void)
{
    material_t m;
    m.color = mk_color(1.0, 1.0, 1.0);
    m.ambient = 0.1;
    m.diffuse = 0.9;
    m.specular = 0.9;
    m.shininess = 200.0;
    m.pattern = pattern_new(ORC_SOLID, mk_color(1.0, 1.0, 1.0), mk_color(0.0, 0.0, 0.0));
    m.reflective = 0.0;
    m.transparency = 0.0;
    m.refractive_index = 1.0;
    m.casts_shadow = 1;
    return m;
}
/* objects.rs:37-52 */
static void object_init(orc_object *o, int shape, double ymin, double ymax, int closed)
{
    static const double list0[15] = { 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    memset(o, 0, sizeof *o);
    o->material = material_new();
    o->transform = m4_identity();
    o->inverse = m4_identity();
    o->inverse_t = m4_identity();
    o->shape = shape;
    if (shape == ORC_CYLINDER || shape == ORC_CONE) {
        o->ymin = ymin; o->ymax = ymax; o->closed = closed;
    }
    memcpy(o->tlist, list0, sizeof list0);
}
/* objects.rs:211-215 */
static void object_set_transform(orc_object *o, mat4_t t)
{
    o->transform = t;
    o->inverse = m4_invert(t);
    o->inverse_t = m4_transpose(o->inverse);
}
/* traits.rs:8-33: translation * scaling * rotation_x * rotation_y * rotation_z * shear, left to right */
static void object_update_transform(orc_object *o)
{
    const double *l = o->tlist;
    mat4_t t = translation(l[0], l[1], l[2]);
    mat4_t s = scaling(l[3], l[4], l[5]);
    mat4_t rx = rotation_x(l[6]), ry = rotation_y(l[7]), rz = rotation_z(l[8]);
    mat4_t sh = shearing(l[9], l[10], l[11], l[12], l[13], l[14]);
    mat4_t m = m4_mul(m4_mul(m4_mul(m4_mul(m4_mul(t, s), rx), ry), rz), sh);
    object_set_transform(o, m);
}

/* derived PartialEq chains: patterns.rs:3, materials.rs:3, objects.rs:3,13 */
static int pattern_eq(const pattern_t *a, const pattern_t *b)
{
    return a->type == b->type && m4_eq(&a->transform, &b->transform) && m4_eq(&a->inverse, &b->inverse) &&
           color_eq(a->c1, b->c1) && color_eq(a->c2, b->c2);
}
static int material_eq(const material_t *a, const material_t *b)
{
    return color_eq(a->color, b->color) && a->ambient == b->ambient && a->diffuse == b->diffuse &&
           a->specular == b->specular && a->shininess == b->shininess && pattern_eq(&a->pattern, &b->pattern) &&
           a->reflective == b->reflective && a->transparency == b->transparency &&
           a->refractive_index == b->refractive_index && a->casts_shadow == b->casts_shadow;
}
static int shape_eq(const orc_object *a, const orc_object *b)
{
    if (a->shape != b->shape) return 0;
    if (a->shape == ORC_CYLINDER || a->shape == ORC_CONE)
        return a->ymin == b->ymin && a->ymax == b->ymax && a->closed == b->closed;
    return 1;
}
static int object_eq(const orc_object *a, const orc_object *b)
{
    if (!material_eq(&a->material, &b->material)) return 0;
    if (!m4_eq(&a->transform, &b->transform) || !m4_eq(&a->inverse, &b->inverse) || !m4_eq(&a->inverse_t, &b->inverse_t))
        return 0;
    if (!shape_eq(a, b)) return 0;
    for (int i = 0; i < 15; i++)
        if (a->tlist[i] != b->tlist[i]) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------
 * rays and per-shape intersection (rays.rs and the five shape modules)
 * ---------------------------------------------------------------------------------- */
typedef struct { tuple_t origin, direction; } ray_t;

/* rays.rs:11-22 */
static ray_t ray_new(tuple_t origin, tuple_t direction)
{
    if (origin.v != 1) panic("Attempted to create ray with vector origin");
    if (direction.v != 0) panic("Attempted to create ray with point direction");
    ray_t r = { origin, direction };
    return r;
}
/* rays.rs:24-28 */
static tuple_t ray_position(ray_t r, double t) { return t_add(r.origin, t_scale(r.direction, t)); }
/* rays.rs:30-34 */
static ray_t ray_transform(ray_t r, const mat4_t *m)
{
    return ray_new(m4_mul_tuple(m, r.origin), m4_mul_tuple(m, r.direction));
}

/* spheres.rs:24-42 */
static int sphere_intersect(ray_t ray, double *ts)
{
    tuple_t v = t_sub(ray.origin, point(0.0, 0.0, 0.0));
    double a = dot(ray.direction, ray.direction);
    double b = 2.0 * dot(ray.direction, v);
    double c = dot(v, v) - 1.0;
    double discriminant = b * b - (4.0 * a * c);
    if (discriminant < 0.0) return 0;
    double t1 = (-b - sqrt(discriminant)) / (2.0 * a);
    double t2 = (-b + sqrt(discriminant)) / (2.0 * a);
    if (t1 < t2) { ts[0] = t1; ts[1] = t2; } else { ts[0] = t2; ts[1] = t1; }
    return 2;
}
/* planes.rs:11-17 */
static int plane_intersect(ray_t ray, double *ts)
{
    if (approx_eq(ray.direction.y, 0.0)) return 0;
    ts[0] = -ray.origin.y / ray.direction.y;
    return 1;
}
/* cubes.rs:31-51 */
static void check_axis(double origin, double direction, double *tmin_out, double *tmax_out)
{
    double tmin_numerator = -1.0 - origin;
    double tmax_numerator = 1.0 - origin;
    double tmin, tmax;
    if (!approx_eq(direction, 0.0)) {
        tmin = tmin_numerator / direction;
        tmax = tmax_numerator / direction;
    } else {
        tmin = tmin_numerator * DBL_MAX;
        tmax = tmax_numerator * DBL_MAX;
    }
    if (tmin > tmax) { double tmp = tmin; tmin = tmax; tmax = tmp; }
    *tmin_out = tmin;
    *tmax_out = tmax;
}
/* cubes.rs:7-29; f64::max/min ignore a NaN operand, as fmax/fmin do */
static int cube_intersect(ray_t ray, double *ts)
{
    double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
    check_axis(ray.origin.x, ray.direction.x, &xtmin, &xtmax);
    check_axis(ray.origin.y, ray.direction.y, &ytmin, &ytmax);
    if (ytmin > xtmax || xtmin > ytmax) return 0;
    check_axis(ray.origin.z, ray.direction.z, &ztmin, &ztmax);
    double tmin = fmax(fmax(xtmin, ytmin), ztmin);
    double tmax = fmin(fmin(xtmax, ytmax), ztmax);
    if (tmin > tmax) return 0;
    ts[0] = tmin;
    ts[1] = tmax;
    return 2;
}
/* cylinders.rs:82-86 */
static int cyl_check_cap(ray_t ray, double t)
{
    double x = ray.origin.x + (t * ray.direction.x);
    double z = ray.origin.z + (t * ray.direction.z);
    return x * x + z * z <= 1.0;
}
/* cylinders.rs:66-80 */
static int cyl_intersect_caps(const orc_object *o, ray_t ray, double *ts, int n)
{
    if (!o->closed || approx_eq(ray.direction.y, 0.0)) return n;
    double t1 = (o->ymin - ray.origin.y) / ray.direction.y;
    if (cyl_check_cap(ray, t1)) ts[n++] = t1;
    double t2 = (o->ymax - ray.origin.y) / ray.direction.y;
    if (cyl_check_cap(ray, t2)) ts[n++] = t2;
    return n;
}
/* cylinders.rs:27-64 */
static int cylinder_intersect(const orc_object *o, ray_t ray, double *ts)
{
    int n = 0;
    double a = ray.direction.x * ray.direction.x + ray.direction.z * ray.direction.z;
    if (a < EPSILON) return cyl_intersect_caps(o, ray, ts, n);
    double b = 2.0 * ray.origin.x * ray.direction.x + 2.0 * ray.origin.z * ray.direction.z;
    double c = ray.origin.x * ray.origin.x + ray.origin.z * ray.origin.z - 1.0;
    double disc = b * b - 4.0 * a * c;
    if (disc < 0.0) return 0;
    double t0 = (-b - sqrt(disc)) / (2.0 * a);
    double t1 = (-b + sqrt(disc)) / (2.0 * a);
    if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
    double y0 = ray.origin.y + (t0 * ray.direction.y);
    if (o->ymin < y0 && y0 < o->ymax) ts[n++] = t0;
    double y1 = ray.origin.y + (t1 * ray.direction.y);
    if (o->ymin < y1 && y1 < o->ymax) ts[n++] = t1;
    return cyl_intersect_caps(o, ray, ts, n);
}
/* cones.rs:100-104: the cap radius test is against |y|, not y^2 */
static int cone_check_cap(ray_t ray, double t, double y)
{
    double x = ray.origin.x + (t * ray.direction.x);
    double z = ray.origin.z + (t * ray.direction.z);
    return x * x + z * z <= fabs(y);
}
/* cones.rs:84-98 */
static int cone_intersect_caps(const orc_object *o, ray_t ray, double *ts, int n)
{
    if (!o->closed || approx_eq(ray.direction.y, 0.0)) return n;
    double t1 = (o->ymin - ray.origin.y) / ray.direction.y;
    if (cone_check_cap(ray, t1, o->ymin)) ts[n++] = t1;
    double t2 = (o->ymax - ray.origin.y) / ray.direction.y;
    if (cone_check_cap(ray, t2, o->ymax)) ts[n++] = t2;
    return n;
}
/* cones.rs:27-82 */
static int cone_intersect(const orc_object *o, ray_t ray, double *ts)
{
    int n = 0;
    double a = ray.direction.x * ray.direction.x - ray.direction.y * ray.direction.y + ray.direction.z * ray.direction.z;
    double b = 2.0 * ray.origin.x * ray.direction.x - 2.0 * ray.origin.y * ray.direction.y +
               2.0 * ray.origin.z * ray.direction.z;
    if (approx_eq(a, 0.0) && approx_eq(b, 0.0)) return cone_intersect_caps(o, ray, ts, n);
    double c = ray.origin.x * ray.origin.x - ray.origin.y * ray.origin.y + ray.origin.z * ray.origin.z;
    if (approx_eq(a, 0.0)) {
        double t = -c / (2.0 * b);
        ts[n++] = t;
        return cone_intersect_caps(o, ray, ts, n);
    }
    double disc = b * b - 4.0 * a * c;
    if (disc < 0.0) return 0;
    double t0 = (-b - sqrt(disc)) / (2.0 * a);
    double t1 = (-b + sqrt(disc)) / (2.0 * a);
    if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
    double y0 = ray.origin.y + (t0 * ray.direction.y);
    if (o->ymin < y0 && y0 < o->ymax) ts[n++] = t0;
    double y1 = ray.origin.y + (t1 * ray.direction.y);
    if (o->ymin < y1 && y1 < o->ymax) ts[n++] = t1;
    return cone_intersect_caps(o, ray, ts, n);
}
/* objects.rs:190-198 */
static int object_local_intersect(const orc_object *o, ray_t local_ray, double *ts)
{
    switch (o->shape) {
    case ORC_SPHERE: return sphere_intersect(local_ray, ts);
    case ORC_PLANE: return plane_intersect(local_ray, ts);
    case ORC_CUBE: return cube_intersect(local_ray, ts);
    case ORC_CYLINDER: return cylinder_intersect(o, local_ray, ts);
    default: return cone_intersect(o, local_ray, ts);
    }
}
/* objects.rs:185-188 */
static int object_intersect(const orc_object *o, ray_t ray, double *ts)
{
    ray_t local_ray = ray_transform(ray, &o->inverse);
    return object_local_intersect(o, local_ray, ts);
}

/* per-shape normals: spheres.rs:20-22, planes.rs:7-9, cubes.rs:53-63, cylinders.rs:96-105, cones.rs:114-127 */
static tuple_t local_normal_at(const orc_object *o, tuple_t p)
{
    switch (o->shape) {
    case ORC_SPHERE: return t_sub(p, point(0.0, 0.0, 0.0));
    case ORC_PLANE: return vector(0.0, 1.0, 0.0);
    case ORC_CUBE: {
        double maxc = fmax(fmax(fabs(p.x), fabs(p.y)), fabs(p.z));
        if (maxc == fabs(p.x)) return vector(p.x, 0.0, 0.0);
        else if (maxc == fabs(p.y)) return vector(0.0, p.y, 0.0);
        else return vector(0.0, 0.0, p.z);
    }
    case ORC_CYLINDER: {
        double d2 = p.x * p.x + p.z * p.z;
        if (d2 < 1.0 && p.y >= o->ymax - EPSILON) return vector(0.0, 1.0, 0.0);
        else if (d2 < 1.0 && p.y <= o->ymin + EPSILON) return vector(0.0, -1.0, 0.0);
        else return vector(p.x, 0.0, p.z);
    }
    default: {
        double d2 = p.x * p.x + p.z * p.z;
        if (d2 < 1.0 && p.y >= o->ymax - EPSILON) return vector(0.0, 1.0, 0.0);
        else if (d2 < 1.0 && p.y <= o->ymin + EPSILON) return vector(0.0, -1.0, 0.0);
        else {
            double y = sqrt(p.x * p.x + p.z * p.z);
            if (p.y > 0.0) y = -y;
            return vector(p.x, y, p.z);
        }
    }
    }
}
/* objects.rs:172-183 */
static tuple_t object_normal_at(const orc_object *o, tuple_t pt)
{
    tuple_t local_point = m4_mul_tuple(&o->inverse, pt);
    tuple_t local_normal = local_normal_at(o, local_point);
    tuple_t world_normal = m4_mul_tuple(&o->inverse_t, local_normal);
    world_normal.v = 0; /* vectorize() */
    return normalize(world_normal);
}

/* Rust `as isize` / `as usize` / `as u8`: saturating, NaN -> 0 */
static long long cast_isize(double x)
{
    if (x != x) return 0;
    if (x >= 9223372036854775807.0) return 9223372036854775807LL;
    if (x <= -9223372036854775808.0) return (-9223372036854775807LL - 1);
    return (long long)x;
}
static unsigned long long cast_usize(double x)
{
    if (x != x || x <= 0.0) return 0;
    if (x >= 18446744073709551615.0) return 18446744073709551615ULL;
    return (unsigned long long)x;
}

/* patterns.rs:94-103, 144-172 */
static color_t pattern_at(const pattern_t *p, tuple_t pt)
{
    switch (p->type) {
    case ORC_STRIPE: return (cast_isize(floor(pt.x)) % 2 == 0) ? p->c1 : p->c2;
    case ORC_GRADIENT: {
        color_t distance = c_sub(p->c2, p->c1);
        double fraction = pt.x - floor(pt.x);
        return c_add(p->c1, c_scale(distance, fraction));
    }
    case ORC_RING: return (cast_usize(floor(sqrt(pt.x * pt.x + pt.z * pt.z))) % 2 == 0) ? p->c1 : p->c2;
    case ORC_CHECKER3D: return (cast_isize(floor(pt.x) + floor(pt.y) + floor(pt.z)) % 2 == 0) ? p->c1 : p->c2;
    case ORC_SOLID: return mk_color(0.0, 0.0, 0.0);
    default: return mk_color(pt.x, pt.y, pt.z);
    }
}
/* objects.rs:200-207 */
static color_t pattern_at_object(const orc_object *o, tuple_t pt)
{
    if (o->material.pattern.type == ORC_SOLID) return o->material.color;
    tuple_t local_point = m4_mul_tuple(&o->inverse, pt);
    tuple_t pattern_point = m4_mul_tuple(&o->material.pattern.inverse, local_point);
    return pattern_at(&o->material.pattern, pattern_point);
}

/* ------------------------------------------------------------------------------------
 * lights (scenes/lights.rs)
 * ---------------------------------------------------------------------------------- */
typedef struct { tuple_t position; color_t intensity; } light_t;

/* lights.rs:43-78 */
static color_t lighting(const material_t *material, const orc_object *object, light_t light, tuple_t pt,
                        tuple_t eyev, tuple_t normalv, int in_shadow)
{
    color_t clr = pattern_at_object(object, pt);
    color_t effective_color = c_mul(clr, light.intensity);
    tuple_t lightv = normalize(t_sub(light.position, pt));
    color_t ambient = c_scale(effective_color, material->ambient);
    if (in_shadow) return ambient;
    color_t diffuse = mk_color(0.0, 0.0, 0.0);
    color_t specular = mk_color(0.0, 0.0, 0.0);
    double light_dot_normal = dot(lightv, normalv);
    if (light_dot_normal >= 0.0) {
        diffuse = c_scale(c_scale(effective_color, material->diffuse), light_dot_normal);
        tuple_t reflectv = t_neg(reflect_vector(lightv, normalv));
        double reflect_dot_eye = dot(reflectv, eyev);
        if (reflect_dot_eye <= 0.0) {
            specular = mk_color(0.0, 0.0, 0.0);
        } else {
            double factor = pow(reflect_dot_eye, material->shininess);
            specular = c_scale(c_scale(light.intensity, material->specular), factor);
        }
    }
    return c_add(c_add(ambient, diffuse), specular);
}

/* ------------------------------------------------------------------------------------
 * world (scenes/world.rs)
 * ---------------------------------------------------------------------------------- */
/* An Object that sits inside Groups (groups.rs:68-84).  The ray reaches it through the inverse
 * of every enclosing group, outermost first, one Ray::transform per level; the Object handed
 * back in its Intersections is a copy whose transformation list was merged with each group's
 * list (combine_transforms, traits.rs:35-44), innermost group first.  That copy does not depend
 * on the ray, so it is made once when the group is added: it is what World.objects[i] holds for
 * such a leaf, while `base` keeps the child as it was added, for the intersection itself. */
#define ORC_MAX_GROUP_DEPTH 8
typedef struct {
    int depth; /* 0: a plain Object of the World */
    mat4_t inverse[ORC_MAX_GROUP_DEPTH];
    orc_object base;
} chain_t;

struct orc_world {
    orc_object *objects;
    chain_t *chains; /* parallel to objects */
    int n_objects, cap_objects;
    light_t *lights;
    int n_lights, cap_lights;
};

/* rays.rs:45-49 holds a copy of the Object; a pointer plus value-equality is the same thing */
typedef struct { double t; const orc_object *object; } isect_t;

typedef struct {
    isect_t *xs; /* scratch for intersect_world; dead once prepare_computations has run */
    isect_t *tmp;
    size_t cap;
    uint64_t n_color, n_shadow;
} trace_ctx;

static void ctx_reserve(trace_ctx *ctx, size_t n)
{
    if (n <= ctx->cap) return;
    size_t cap = ctx->cap ? ctx->cap : 64;
    while (cap < n) cap *= 2;
    ctx->xs = (isect_t *)realloc(ctx->xs, cap * sizeof(isect_t));
    ctx->tmp = (isect_t *)realloc(ctx->tmp, cap * sizeof(isect_t));
    if (!ctx->xs || !ctx->tmp) { fprintf(stderr, "rt_oracle: out of memory\n"); abort(); }
    ctx->cap = cap;
}
static void ctx_free(trace_ctx *ctx) { free(ctx->xs); free(ctx->tmp); memset(ctx, 0, sizeof *ctx); }

/* world.rs:105: stable sort by t; partial_cmp().unwrap() panics on NaN */
static void stable_sort(trace_ctx *ctx, size_t n)
{
    isect_t *a = ctx->xs;
    for (size_t i = 0; i < n; i++)
        if (a[i].t != a[i].t) panic("called `Option::unwrap()` on a `None` value (NaN t in sort)");
    if (n <= 16) { /* insertion sort is stable */
        for (size_t i = 1; i < n; i++) {
            isect_t key = a[i];
            size_t j = i;
            while (j > 0 && a[j - 1].t > key.t) { a[j] = a[j - 1]; j--; }
            a[j] = key;
        }
        return;
    }
    /* bottom-up merge sort, stable: ties take the left run first */
    isect_t *src = a, *dst = ctx->tmp;
    for (size_t width = 1; width < n; width *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * width) {
            size_t mid = lo + width < n ? lo + width : n;
            size_t hi = lo + 2 * width < n ? lo + 2 * width : n;
            size_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) dst[k++] = (src[j].t < src[i].t) ? src[j++] : src[i++];
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        isect_t *sw = src; src = dst; dst = sw;
    }
    if (src != a) memcpy(a, src, n * sizeof(isect_t));
}

/* world.rs:98-107 */
static size_t intersect_world(const orc_world *w, ray_t ray, trace_ctx *ctx)
{
    size_t n = 0;
    for (int i = 0; i < w->n_objects; i++) {
        double ts[4];
        int k;
        const chain_t *ch = &w->chains[i];
        if (ch->depth == 0) {
            k = object_intersect(&w->objects[i], ray, ts);
        } else { /* Group::intersect, groups.rs:68-71, level by level, then the child's own intersect */
            ray_t local = ray;
            for (int g = 0; g < ch->depth; g++) local = ray_transform(local, &ch->inverse[g]);
            k = object_intersect(&ch->base, local, ts);
        }
        ctx_reserve(ctx, n + 4);
        for (int j = 0; j < k; j++) {
            ctx->xs[n].t = ts[j];
            ctx->xs[n].object = &w->objects[i];
            n++;
        }
    }
    stable_sort(ctx, n);
    return n;
}
/* world.rs:109-119 */
static long hit_world(const isect_t *xs, size_t n)
{
    for (size_t i = 0; i < n; i++)
        if (xs[i].t >= 0.0) return (long)i;
    return -1;
}

/* world.rs:195-208 */
typedef struct {
    const orc_object *object;
    double t_value;
    tuple_t point, eyev, normalv;
    int inside;
    tuple_t over_point, under_point, reflectv;
    double n1, n2;
} comps_t;

/* rays.rs:45 derived PartialEq: t == t and object == object */
static int isect_eq(const isect_t *a, const isect_t *b) { return a->t == b->t && object_eq(a->object, b->object); }

/* world.rs:242-287 and Computations::new world.rs:211-225 */
static comps_t prepare_computations(isect_t intersection, ray_t ray, const isect_t *list, size_t n_list)
{
    comps_t c;
    tuple_t pt = ray_position(ray, intersection.t);
    tuple_t eyev = t_neg(ray.direction);
    tuple_t normalv = object_normal_at(intersection.object, pt);
    int inside = 0;
    if (dot(normalv, eyev) < 0.0) {
        inside = 1;
        normalv = t_neg(normalv);
    }
    tuple_t reflectv = reflect_vector(ray.direction, normalv);

    double n1 = 1.0, n2 = 1.0;
    /* containers: Vec<Object>, searched and removed by value equality */
    const orc_object **containers = (const orc_object **)malloc((n_list + 1) * sizeof(*containers));
    size_t n_cont = 0;
    for (size_t k = 0; k < n_list; k++) {
        const isect_t *i = &list[k];
        int is_hit = isect_eq(i, &intersection);
        if (is_hit) n1 = (n_cont == 0) ? 1.0 : containers[n_cont - 1]->material.refractive_index;
        int found = 0;
        for (size_t x = 0; x < n_cont; x++) {
            if (object_eq(containers[x], i->object)) {
                memmove(&containers[x], &containers[x + 1], (n_cont - x - 1) * sizeof(*containers));
                n_cont--;
                found = 1;
                break;
            }
        }
        if (!found) containers[n_cont++] = i->object;
        if (is_hit) {
            n2 = (n_cont == 0) ? 1.0 : containers[n_cont - 1]->material.refractive_index;
            break;
        }
    }
    free(containers);

    c.object = intersection.object;
    c.t_value = intersection.t;
    c.point = pt;
    c.eyev = eyev;
    c.normalv = normalv;
    c.inside = inside;
    c.over_point = t_add(pt, t_scale(normalv, EPSILON));
    c.under_point = t_sub(pt, t_scale(normalv, EPSILON));
    c.reflectv = reflectv;
    c.n1 = n1;
    c.n2 = n2;
    return c;
}

/* world.rs:227-239; powi(2) = x*x, powi(5) = x * (x^2)^2 */
static double schlick(const comps_t *c)
{
    double cosv = dot(c->eyev, c->normalv);
    if (c->n1 > c->n2) {
        double n = c->n1 / c->n2;
        double sin2_t = (n * n) * (1.0 - (cosv * cosv));
        if (sin2_t > 1.0) return 1.0;
        cosv = sqrt(1.0 - sin2_t);
    }
    double q = (c->n1 - c->n2) / (c->n1 + c->n2);
    double r0 = q * q;
    double x = 1.0 - cosv;
    double x2 = x * x;
    double x4 = x2 * x2;
    return r0 + (1.0 - r0) * (x * x4);
}

static color_t color_at(const orc_world *w, ray_t ray, uint32_t remaining, trace_ctx *ctx);

/* world.rs:146-158 */
static int is_shadowed(const orc_world *w, tuple_t pt, trace_ctx *ctx)
{
    ctx->n_shadow++;
    if (w->n_lights == 0) panic("index out of bounds: the len is 0 but the index is 0 (World.lights)");
    light_t light = w->lights[0];
    tuple_t v = t_sub(light.position, pt);
    double distance = magnitude(v);
    ray_t ray = ray_new(pt, normalize(v));
    size_t n = intersect_world(w, ray, ctx);
    long h = hit_world(ctx->xs, n);
    if (h >= 0) {
        if (ctx->xs[h].t < distance && ctx->xs[h].object->material.casts_shadow) return 1;
    }
    return 0;
}
/* world.rs:160-173 */
static color_t reflected_color(const orc_world *w, const comps_t *c, uint32_t remaining, trace_ctx *ctx)
{
    if (remaining < 1) return mk_color(0.0, 0.0, 0.0);
    double reflective = c->object->material.reflective;
    if (reflective == 0.0) return mk_color(0.0, 0.0, 0.0);
    ray_t ray = ray_new(c->over_point, c->reflectv);
    color_t clr = color_at(w, ray, remaining - 1, ctx);
    return c_scale(clr, reflective);
}
/* world.rs:175-190 */
static color_t refracted_color(const orc_world *w, const comps_t *c, uint32_t remaining, trace_ctx *ctx)
{
    if (c->object->material.transparency == 0.0 || remaining == 0) return mk_color(0.0, 0.0, 0.0);
    double n_ratio = c->n1 / c->n2;
    double cos_i = dot(c->eyev, c->normalv);
    double sin2_t = (n_ratio * n_ratio) * (1.0 - (cos_i * cos_i));
    if (sin2_t > 1.0) return mk_color(0.0, 0.0, 0.0);
    double cos_t = sqrt(1.0 - sin2_t);
    tuple_t direction = t_sub(t_scale(c->normalv, (n_ratio * cos_i) - cos_t), t_scale(c->eyev, n_ratio));
    ray_t refract_ray = ray_new(c->under_point, direction);
    return c_scale(color_at(w, refract_ray, remaining - 1, ctx), c->object->material.transparency);
}
/* world.rs:121-134 */
static color_t shade_hit(const orc_world *w, const comps_t *c, uint32_t remaining, trace_ctx *ctx)
{
    int shadowed = is_shadowed(w, c->over_point, ctx);
    color_t clr = lighting(&c->object->material, c->object, w->lights[0], c->over_point, c->eyev, c->normalv, shadowed);
    color_t reflections = reflected_color(w, c, remaining, ctx);
    color_t refractions = refracted_color(w, c, remaining, ctx);
    if (c->object->material.reflective > 0.0 && c->object->material.transparency > 0.0) {
        double reflectance = schlick(c);
        return c_add(c_add(clr, c_scale(reflections, reflectance)), c_scale(refractions, 1.0 - reflectance));
    }
    return c_add(c_add(clr, reflections), refractions);
}
/* world.rs:136-144 */
static color_t color_at(const orc_world *w, ray_t ray, uint32_t remaining, trace_ctx *ctx)
{
    ctx->n_color++;
    size_t n = intersect_world(w, ray, ctx);
    long h = hit_world(ctx->xs, n);
    if (h < 0) return mk_color(0.0, 0.0, 0.0);
    comps_t c = prepare_computations(ctx->xs[h], ray, ctx->xs, n);
    return shade_hit(w, &c, remaining, ctx);
}

/* ------------------------------------------------------------------------------------
 * camera (scenes/camera.rs)
 * ---------------------------------------------------------------------------------- */
struct orc_camera {
    uint64_t hsize, vsize;
    double field_of_view;
    mat4_t transform, inverse;
    tuple_t from, to, up; /* transform_components */
    double pixel_size, half_width, half_height;
};

/* camera.rs:87-99 */
static void camera_initialize(orc_camera *c)
{
    double half_view = tan(c->field_of_view / 2.0);
    double aspect = (double)c->hsize / (double)c->vsize;
    if (aspect >= 1.0) {
        c->half_width = half_view;
        c->half_height = half_view / aspect;
    } else {
        c->half_width = half_view * aspect;
        c->half_height = half_view;
    }
    c->pixel_size = (c->half_width * 2.0) / (double)c->hsize;
}
/* camera.rs:101-105 */
static void camera_set_transform(orc_camera *c, mat4_t t)
{
    c->transform = t;
    c->inverse = m4_invert(t);
}
/* camera.rs:138-145 */
static void camera_update_transformations(orc_camera *c) { camera_set_transform(c, view_transform(c->from, c->to, c->up)); }
/* camera.rs:147-158 */
static ray_t ray_for_pixel(const orc_camera *c, uint64_t x, uint64_t y)
{
    double xoffset = ((double)x + 0.5) * c->pixel_size;
    double yoffset = ((double)y + 0.5) * c->pixel_size;
    double world_x = c->half_width - xoffset;
    double world_y = c->half_height - yoffset;
    tuple_t pixel = m4_mul_tuple(&c->inverse, point(world_x, world_y, -1.0));
    tuple_t origin = m4_mul_tuple(&c->inverse, point(0.0, 0.0, 0.0));
    tuple_t direction = normalize(t_sub(pixel, origin));
    return ray_new(origin, direction);
}

/* ====================================================================================
 * exported API
 * ================================================================================== */
static mat4_t m4_load(const double *a)
{
    mat4_t r;
    memcpy(r.m, a, sizeof r.m);
    return r;
}
static void m4_store(mat4_t m, double *out) { memcpy(out, m.m, sizeof m.m); }

void orc_identity(double *out) { m4_store(m4_identity(), out); }
void orc_translation(double x, double y, double z, double *out) { m4_store(translation(x, y, z), out); }
void orc_scaling(double x, double y, double z, double *out) { m4_store(scaling(x, y, z), out); }
void orc_rotation_x(double rad, double *out) { m4_store(rotation_x(rad), out); }
void orc_rotation_y(double rad, double *out) { m4_store(rotation_y(rad), out); }
void orc_rotation_z(double rad, double *out) { m4_store(rotation_z(rad), out); }
void orc_shearing(double xy, double xz, double yx, double yz, double zx, double zy, double *out)
{
    m4_store(shearing(xy, xz, yx, yz, zx, zy), out);
}
void orc_mat4_mul(const double *a, const double *b, double *out) { m4_store(m4_mul(m4_load(a), m4_load(b)), out); }
void orc_mat4_transpose(const double *a, double *out) { m4_store(m4_transpose(m4_load(a)), out); }
double orc_mat4_determinant(const double *a)
{
    mat4_t m = m4_load(a);
    return det4(&m);
}
int orc_mat4_invert(const double *a, double *out)
{
    API_ENTER();
    m4_store(m4_invert(m4_load(a)), out);
    API_LEAVE();
    return ORC_OK;
}
void orc_mat4_mul_tuple(const double *m, const double *tup4, double *out4)
{
    mat4_t a = m4_load(m);
    tuple_t t = { tup4[0], tup4[1], tup4[2], (int)tup4[3] };
    tuple_t r = m4_mul_tuple(&a, t);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = (double)r.v;
}
int orc_view_transform(const double *f, const double *t, const double *u, double *out)
{
    API_ENTER();
    m4_store(view_transform(point(f[0], f[1], f[2]), point(t[0], t[1], t[2]), vector(u[0], u[1], u[2])), out);
    API_LEAVE();
    return ORC_OK;
}
int orc_normalize(const double *v3, double *out3)
{
    API_ENTER();
    tuple_t r = normalize(vector(v3[0], v3[1], v3[2]));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
    API_LEAVE();
    return ORC_OK;
}
void orc_reflect_vector(const double *v3, const double *n3, double *out3)
{
    tuple_t r = reflect_vector(vector(v3[0], v3[1], v3[2]), vector(n3[0], n3[1], n3[2]));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

orc_object *orc_object_new(int shape, double ymin, double ymax, int closed)
{
    orc_object *o = (orc_object *)malloc(sizeof *o);
    object_init(o, shape, ymin, ymax, closed);
    return o;
}
/* spheres.rs:7-19 */
orc_object *orc_glass_sphere(void)
{
    orc_object *o = orc_object_new(ORC_SPHERE, 0, 0, 0);
    o->material.transparency = 0.9;
    o->material.reflective = 0.9;
    o->material.refractive_index = 1.5;
    o->material.color = mk_color(0.0, 0.0, 0.0);
    o->material.casts_shadow = 0;
    o->material.ambient = 0.0;
    o->material.diffuse = 0.0;
    o->material.shininess = 300.0;
    o->material.specular = 0.9;
    return o;
}
orc_object *orc_object_clone(const orc_object *o)
{
    orc_object *c = (orc_object *)malloc(sizeof *c);
    *c = *o;
    return c;
}
void orc_object_free(orc_object *o) { free(o); }
int orc_object_set_transform(orc_object *o, const double *m16)
{
    API_ENTER();
    object_set_transform(o, m4_load(m16));
    API_LEAVE();
    return ORC_OK;
}
/* objects.rs:217-220 */
int orc_object_set_transformation_list(orc_object *o, int index, double x)
{
    API_ENTER();
    if (index < 0 || index >= 15) panic("index out of bounds (transformations_list)");
    o->tlist[index] = x;
    object_update_transform(o);
    API_LEAVE();
    return ORC_OK;
}
/* traits.rs:46-98: scale_x/y/z replace 0 by EPSILON */
int orc_object_transform_op(orc_object *o, int slot, double x)
{
    if (slot >= 3 && slot <= 5 && !(x != 0.0)) x = EPSILON;
    return orc_object_set_transformation_list(o, slot, x);
}
void orc_object_set_color(orc_object *o, double r, double g, double b) { o->material.color = mk_color(r, g, b); }
void orc_object_set_scalar(orc_object *o, int field, double value)
{
    switch (field) {
    case ORC_AMBIENT: o->material.ambient = value; break;
    case ORC_DIFFUSE: o->material.diffuse = value; break;
    case ORC_SPECULAR: o->material.specular = value; break;
    case ORC_SHININESS: o->material.shininess = value; break;
    case ORC_REFLECTIVE: o->material.reflective = value; break;
    case ORC_TRANSPARENCY: o->material.transparency = value; break;
    case ORC_REFRACTIVE_INDEX: o->material.refractive_index = value; break;
    default: break;
    }
}
void orc_object_set_casts_shadow(orc_object *o, int casts) { o->material.casts_shadow = casts ? 1 : 0; }
void orc_object_set_pattern(orc_object *o, int type, const double *c1, const double *c2)
{
    o->material.pattern = pattern_new(type, mk_color(c1[0], c1[1], c1[2]), mk_color(c2[0], c2[1], c2[2]));
}
/* patterns.rs:88-92 */
int orc_object_set_pattern_transform(orc_object *o, const double *m16)
{
    API_ENTER();
    o->material.pattern.transform = m4_load(m16);
    o->material.pattern.inverse = m4_invert(o->material.pattern.transform);
    API_LEAVE();
    return ORC_OK;
}
void orc_object_get_matrices(const orc_object *o, double *transform, double *inverse, double *inverse_t)
{
    if (transform) m4_store(o->transform, transform);
    if (inverse) m4_store(o->inverse, inverse);
    if (inverse_t) m4_store(o->inverse_t, inverse_t);
}
void orc_object_get_pattern_inverse(const orc_object *o, double *inverse) { m4_store(o->material.pattern.inverse, inverse); }
int orc_object_equal(const orc_object *a, const orc_object *b) { return object_eq(a, b); }
int orc_object_normal_at(const orc_object *o, const double *p3, double *out3)
{
    API_ENTER();
    tuple_t n = object_normal_at(o, point(p3[0], p3[1], p3[2]));
    out3[0] = n.x; out3[1] = n.y; out3[2] = n.z;
    API_LEAVE();
    return ORC_OK;
}
static ray_t ray_load(const double *r) { return ray_new(point(r[0], r[1], r[2]), vector(r[3], r[4], r[5])); }
int orc_object_intersect(const orc_object *o, const double *ray6, double *out_t) { return object_intersect(o, ray_load(ray6), out_t); }
int orc_object_local_intersect(const orc_object *o, const double *ray6, double *out_t)
{
    return object_local_intersect(o, ray_load(ray6), out_t);
}
void orc_object_pattern_at_object(const orc_object *o, const double *p3, double *out3)
{
    color_t c = pattern_at_object(o, point(p3[0], p3[1], p3[2]));
    out3[0] = c.r; out3[1] = c.g; out3[2] = c.b;
}
void orc_object_pattern_at(const orc_object *o, const double *p3, double *out3)
{
    color_t c = pattern_at(&o->material.pattern, point(p3[0], p3[1], p3[2]));
    out3[0] = c.r; out3[1] = c.g; out3[2] = c.b;
}

int orc_lighting(const orc_object *o, const double *lp, const double *li, const double *p, const double *e,
                 const double *n, int in_shadow, double *out3)
{
    API_ENTER();
    light_t light = { point(lp[0], lp[1], lp[2]), mk_color(li[0], li[1], li[2]) };
    color_t c = lighting(&o->material, o, light, point(p[0], p[1], p[2]), vector(e[0], e[1], e[2]),
                         vector(n[0], n[1], n[2]), in_shadow);
    out3[0] = c.r; out3[1] = c.g; out3[2] = c.b;
    API_LEAVE();
    return ORC_OK;
}

orc_world *orc_world_new(void) { return (orc_world *)calloc(1, sizeof(orc_world)); }
/* world.rs:38-66 */
orc_world *orc_world_default(void)
{
    orc_world *w = orc_world_new();
    orc_object s1, s2;
    object_init(&s1, ORC_SPHERE, 0, 0, 0);
    s1.material.color = mk_color(0.8, 1.0, 0.6);
    s1.material.ambient = 0.1;
    s1.material.diffuse = 0.7;
    s1.material.specular = 0.2;
    s1.material.shininess = 200.0;
    object_init(&s2, ORC_SPHERE, 0, 0, 0);
    orc_object_transform_op(&s2, 3, 0.5);
    orc_object_transform_op(&s2, 4, 0.5);
    orc_object_transform_op(&s2, 5, 0.5);
    orc_world_add_object(w, &s1);
    orc_world_add_object(w, &s2);
    double pos[3] = { -10.0, 10.0, -10.0 }, in[3] = { 1.0, 1.0, 1.0 };
    orc_world_add_light(w, pos, in);
    return w;
}
void orc_world_free(orc_world *w)
{
    if (!w) return;
    free(w->objects);
    free(w->chains);
    free(w->lights);
    free(w);
}
static int world_push(orc_world *w, const orc_object *o)
{
    if (w->n_objects == w->cap_objects) {
        w->cap_objects = w->cap_objects ? 2 * w->cap_objects : 16;
        w->objects = (orc_object *)realloc(w->objects, (size_t)w->cap_objects * sizeof(orc_object));
        w->chains = (chain_t *)realloc(w->chains, (size_t)w->cap_objects * sizeof(chain_t));
        if (!w->objects || !w->chains) { fprintf(stderr, "rt_oracle: out of memory\n"); abort(); }
    }
    w->objects[w->n_objects] = *o;
    w->chains[w->n_objects].depth = 0;
    return w->n_objects++;
}
int orc_world_add_object(orc_world *w, const orc_object *o) { return world_push(w, o); }

/* ---- groups (groups.rs, objectholders.rs) ---- */
typedef struct {
    int is_group;
    orc_object object;
    orc_group *group;
} holder_t;
struct orc_group {
    holder_t *items;
    int n, cap;
    mat4_t transform, inverse, inverse_t;
    double tlist[15];
};
/* groups.rs:27-43 */
orc_group *orc_group_new(void)
{
    orc_group *g = (orc_group *)calloc(1, sizeof(orc_group));
    g->transform = m4_identity();
    g->inverse = m4_identity();
    g->inverse_t = m4_identity();
    g->tlist[3] = g->tlist[4] = g->tlist[5] = 1.0;
    return g;
}
void orc_group_free(orc_group *g)
{
    if (!g) return;
    for (int i = 0; i < g->n; i++)
        if (g->items[i].is_group) orc_group_free(g->items[i].group);
    free(g->items);
    free(g);
}
static orc_group *group_clone(const orc_group *g)
{
    orc_group *c = (orc_group *)malloc(sizeof(orc_group));
    *c = *g;
    c->items = (holder_t *)malloc((size_t)(g->cap ? g->cap : 1) * sizeof(holder_t));
    for (int i = 0; i < g->n; i++) {
        c->items[i] = g->items[i];
        if (g->items[i].is_group) c->items[i].group = group_clone(g->items[i].group);
    }
    return c;
}
static holder_t *group_slot(orc_group *g)
{
    if (g->n == g->cap) {
        g->cap = g->cap ? 2 * g->cap : 8;
        g->items = (holder_t *)realloc(g->items, (size_t)g->cap * sizeof(holder_t));
    }
    return &g->items[g->n++];
}
/* groups.rs:49-55: values are moved in, i.e. the group owns copies */
int orc_group_add_object(orc_group *g, const orc_object *o)
{
    holder_t *h = group_slot(g);
    h->is_group = 0;
    h->object = *o;
    h->group = NULL;
    return g->n - 1;
}
int orc_group_add_group(orc_group *g, const orc_group *child)
{
    holder_t *h = group_slot(g);
    h->is_group = 1;
    h->group = group_clone(child);
    return g->n - 1;
}
int orc_group_n_elements(const orc_group *g) { return g->n; }
size_t orc_object_sizeof(void) { return sizeof(orc_object); }
/* groups.rs:88-92 */
static void group_set_transform(orc_group *g, mat4_t t)
{
    g->transform = t;
    g->inverse = m4_invert(t);
    g->inverse_t = m4_transpose(g->inverse);
}
int orc_group_set_transform(orc_group *g, const double *m16)
{
    API_ENTER();
    group_set_transform(g, m4_load(m16));
    API_LEAVE();
    return ORC_OK;
}
/* groups.rs:94-97 with traits.rs:8-33 */
int orc_group_set_transformation_list(orc_group *g, int index, double x)
{
    API_ENTER();
    if (index < 0 || index >= 15) panic("index out of bounds (transformations_list)");
    g->tlist[index] = x;
    const double *l = g->tlist;
    mat4_t t = translation(l[0], l[1], l[2]);
    mat4_t sc = scaling(l[3], l[4], l[5]);
    mat4_t rx = rotation_x(l[6]), ry = rotation_y(l[7]), rz = rotation_z(l[8]);
    mat4_t sh = shearing(l[9], l[10], l[11], l[12], l[13], l[14]);
    group_set_transform(g, m4_mul(m4_mul(m4_mul(m4_mul(m4_mul(t, sc), rx), ry), rz), sh));
    API_LEAVE();
    return ORC_OK;
}
/* traits.rs:46-98: scale_x/y/z replace 0 by EPSILON */
int orc_group_transform_op(orc_group *g, int slot, double x)
{
    if (slot >= 3 && slot <= 5 && !(x != 0.0)) x = EPSILON;
    return orc_group_set_transformation_list(g, slot, x);
}
void orc_group_get_matrices(const orc_group *g, double *transform, double *inverse, double *inverse_t)
{
    if (transform) m4_store(g->transform, transform);
    if (inverse) m4_store(g->inverse, inverse);
    if (inverse_t) m4_store(g->inverse_t, inverse_t);
}
/* traits.rs:35-44 */
static void object_combine_transforms(orc_object *o, const double *list)
{
    for (int i = 0; i < 15; i++) {
        if (i <= 2 || i >= 6) o->tlist[i] += list[i];
        else o->tlist[i] *= list[i];
    }
    object_update_transform(o);
}
/* Group::intersect + local_intersect, groups.rs:68-84, as written: recursive, every level sorts.
 * Returns the number of intersections appended at out[n0..]; objs receives the merged copies. */
typedef struct { double t; orc_object object; } gisect_t;
static int group_intersect(const orc_group *g, ray_t ray, gisect_t **out, int *cap, int n0)
{
    ray_t local = ray_transform(ray, &g->inverse);
    int n = n0;
    for (int i = 0; i < g->n; i++) {
        int first = n;
        if (g->items[i].is_group) {
            n = group_intersect(g->items[i].group, local, out, cap, n);
        } else {
            double ts[4];
            int k = object_intersect(&g->items[i].object, local, ts);
            if (n + k > *cap) {
                *cap = 2 * (*cap) + 16;
                *out = (gisect_t *)realloc(*out, (size_t)*cap * sizeof(gisect_t));
            }
            for (int j = 0; j < k; j++) {
                (*out)[n].t = ts[j];
                (*out)[n].object = g->items[i].object;
                n++;
            }
        }
        for (int j = first; j < n; j++) object_combine_transforms(&(*out)[j].object, g->tlist);
    }
    /* vec.sort_by(partial_cmp().unwrap()): stable, panics on NaN */
    gisect_t *a = *out + n0;
    int m = n - n0;
    for (int i = 0; i < m; i++)
        if (a[i].t != a[i].t) panic("called `Option::unwrap()` on a `None` value (NaN t in sort)");
    for (int i = 1; i < m; i++) {
        gisect_t key = a[i];
        int j = i;
        while (j > 0 && a[j - 1].t > key.t) { a[j] = a[j - 1]; j--; }
        a[j] = key;
    }
    return n;
}
int orc_group_intersect(const orc_group *g, const double *ray6, double *out_t, orc_object *out_objects, int cap)
{
    gisect_t *xs = NULL;
    int xcap = 0, n = -1;
    jmp_buf jb;
    jmp_buf *saved = tl_jmp;
    if (setjmp(jb)) { tl_jmp = saved; free(xs); return -1; }
    tl_jmp = &jb;
    n = group_intersect(g, ray_load(ray6), &xs, &xcap, 0);
    for (int i = 0; i < n && i < cap; i++) {
        if (out_t) out_t[i] = xs[i].t;
        if (out_objects) out_objects[i] = xs[i].object;
    }
    tl_jmp = saved;
    free(xs);
    return n;
}
/* World::add_group, world.rs:82-84.  The leaves are laid out depth first, which is the order
 * the nested stable sorts leave equal t in. */
static void world_add_leaves(orc_world *w, const orc_group *g, chain_t *chain, const double **lists)
{
    if (chain->depth >= ORC_MAX_GROUP_DEPTH) panic("groups nested deeper than this restatement supports");
    chain->inverse[chain->depth] = g->inverse;
    lists[chain->depth] = g->tlist;
    chain->depth++;
    for (int i = 0; i < g->n; i++) {
        if (g->items[i].is_group) {
            world_add_leaves(w, g->items[i].group, chain, lists);
        } else {
            orc_object merged = g->items[i].object;
            for (int d = chain->depth - 1; d >= 0; d--) object_combine_transforms(&merged, lists[d]);
            int idx = world_push(w, &merged);
            w->chains[idx] = *chain;
            w->chains[idx].base = g->items[i].object;
        }
    }
    chain->depth--;
}
int orc_world_add_group(orc_world *w, const orc_group *g)
{
    API_ENTER();
    chain_t chain;
    const double *lists[ORC_MAX_GROUP_DEPTH];
    memset(&chain, 0, sizeof chain);
    int first = w->n_objects;
    world_add_leaves(w, g, &chain, lists);
    API_LEAVE();
    return first;
}
/* depth of the group chain above World.objects[index] and the inverse of each group, outermost first */
int orc_world_object_chain(const orc_world *w, int index, double *inverses16, int cap)
{
    if (index < 0 || index >= w->n_objects) return -1;
    const chain_t *ch = &w->chains[index];
    for (int d = 0; d < ch->depth && d < cap; d++) m4_store(ch->inverse[d], inverses16 + 16 * d);
    return ch->depth;
}
const orc_object *orc_world_object_base(const orc_world *w, int index)
{
    if (index < 0 || index >= w->n_objects) return NULL;
    return w->chains[index].depth ? &w->chains[index].base : &w->objects[index];
}
void orc_world_add_light(orc_world *w, const double *pos3, const double *in3)
{
    if (w->n_lights == w->cap_lights) {
        w->cap_lights = w->cap_lights ? 2 * w->cap_lights : 4;
        w->lights = (light_t *)realloc(w->lights, (size_t)w->cap_lights * sizeof(light_t));
    }
    w->lights[w->n_lights].position = point(pos3[0], pos3[1], pos3[2]);
    w->lights[w->n_lights].intensity = mk_color(in3[0], in3[1], in3[2]);
    w->n_lights++;
}
void orc_world_clear_lights(orc_world *w) { w->n_lights = 0; }
int orc_world_n_objects(const orc_world *w) { return w->n_objects; }
orc_object *orc_world_object(orc_world *w, int index) { return (index >= 0 && index < w->n_objects) ? &w->objects[index] : NULL; }

int orc_world_intersect(const orc_world *w, const double *ray6, double *out_t, int *out_obj, int cap)
{
    trace_ctx ctx;
    memset(&ctx, 0, sizeof ctx);
    jmp_buf jb;
    jmp_buf *saved = tl_jmp;
    if (setjmp(jb)) { tl_jmp = saved; ctx_free(&ctx); return -1; }
    tl_jmp = &jb;
    size_t n = intersect_world(w, ray_load(ray6), &ctx);
    for (size_t i = 0; i < n && (int)i < cap; i++) {
        out_t[i] = ctx.xs[i].t;
        if (out_obj) out_obj[i] = (int)(ctx.xs[i].object - w->objects);
    }
    tl_jmp = saved;
    ctx_free(&ctx);
    return (int)n;
}
int orc_world_color_at_batch(const orc_world *w, const double *rays6, uint64_t n, uint32_t remaining, double *out_rgb,
                             uint64_t *counters2)
{
    trace_ctx ctx;
    memset(&ctx, 0, sizeof ctx);
    jmp_buf jb;
    jmp_buf *saved = tl_jmp;
    if (setjmp(jb)) { tl_jmp = saved; ctx_free(&ctx); return ORC_PANIC; }
    tl_jmp = &jb;
    for (uint64_t i = 0; i < n; i++) {
        color_t c = color_at(w, ray_load(rays6 + 6 * i), remaining, &ctx);
        out_rgb[3 * i + 0] = c.r; out_rgb[3 * i + 1] = c.g; out_rgb[3 * i + 2] = c.b;
    }
    if (counters2) { counters2[0] += ctx.n_color; counters2[1] += ctx.n_shadow; }
    tl_jmp = saved;
    ctx_free(&ctx);
    return ORC_OK;
}
int orc_world_color_at(const orc_world *w, const double *ray6, uint32_t remaining, double *out3)
{
    return orc_world_color_at_batch(w, ray6, 1, remaining, out3, NULL);
}
int orc_world_is_shadowed(const orc_world *w, const double *p3, int *out)
{
    trace_ctx ctx;
    memset(&ctx, 0, sizeof ctx);
    jmp_buf jb;
    jmp_buf *saved = tl_jmp;
    if (setjmp(jb)) { tl_jmp = saved; ctx_free(&ctx); return ORC_PANIC; }
    tl_jmp = &jb;
    *out = is_shadowed(w, point(p3[0], p3[1], p3[2]), &ctx);
    tl_jmp = saved;
    ctx_free(&ctx);
    return ORC_OK;
}

static void comps_export(const comps_t *c, const orc_object *const *objs, int n, orc_comps *out)
{
    out->t = c->t_value;
    out->point[0] = c->point.x; out->point[1] = c->point.y; out->point[2] = c->point.z;
    out->eyev[0] = c->eyev.x; out->eyev[1] = c->eyev.y; out->eyev[2] = c->eyev.z;
    out->normalv[0] = c->normalv.x; out->normalv[1] = c->normalv.y; out->normalv[2] = c->normalv.z;
    out->over_point[0] = c->over_point.x; out->over_point[1] = c->over_point.y; out->over_point[2] = c->over_point.z;
    out->under_point[0] = c->under_point.x; out->under_point[1] = c->under_point.y; out->under_point[2] = c->under_point.z;
    out->reflectv[0] = c->reflectv.x; out->reflectv[1] = c->reflectv.y; out->reflectv[2] = c->reflectv.z;
    out->n1 = c->n1;
    out->n2 = c->n2;
    out->inside = c->inside;
    out->object_index = -1;
    for (int i = 0; i < n; i++)
        if (objs[i] == c->object) { out->object_index = i; break; }
}
static comps_t comps_import(const orc_comps *c, const orc_object *obj)
{
    comps_t r;
    r.object = obj;
    r.t_value = c->t;
    r.point = point(c->point[0], c->point[1], c->point[2]);
    r.eyev = vector(c->eyev[0], c->eyev[1], c->eyev[2]);
    r.normalv = vector(c->normalv[0], c->normalv[1], c->normalv[2]);
    r.inside = c->inside;
    r.over_point = point(c->over_point[0], c->over_point[1], c->over_point[2]);
    r.under_point = point(c->under_point[0], c->under_point[1], c->under_point[2]);
    r.reflectv = vector(c->reflectv[0], c->reflectv[1], c->reflectv[2]);
    r.n1 = c->n1;
    r.n2 = c->n2;
    return r;
}
int orc_prepare_computations(int n, const double *ts, const orc_object *const *objs, int hit_index, const double *ray6,
                             orc_comps *out)
{
    isect_t *list = (isect_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(isect_t));
    jmp_buf jb;
    jmp_buf *saved = tl_jmp;
    if (setjmp(jb)) { tl_jmp = saved; free(list); return ORC_PANIC; }
    tl_jmp = &jb;
    for (int i = 0; i < n; i++) { list[i].t = ts[i]; list[i].object = objs[i]; }
    comps_t c = prepare_computations(list[hit_index], ray_load(ray6), list, (size_t)n);
    comps_export(&c, objs, n, out);
    out->object_index = hit_index;
    tl_jmp = saved;
    free(list);
    return ORC_OK;
}
double orc_schlick(const orc_comps *c)
{
    comps_t r = comps_import(c, NULL);
    return schlick(&r);
}
#define WORLD_CALL(expr)                                                              \
    trace_ctx ctx;                                                                    \
    memset(&ctx, 0, sizeof ctx);                                                      \
    jmp_buf jb;                                                                       \
    jmp_buf *saved = tl_jmp;                                                          \
    if (setjmp(jb)) { tl_jmp = saved; ctx_free(&ctx); return ORC_PANIC; }             \
    tl_jmp = &jb;                                                                     \
    comps_t r = comps_import(c, obj);                                                 \
    color_t col = expr;                                                               \
    out3[0] = col.r; out3[1] = col.g; out3[2] = col.b;                                \
    tl_jmp = saved;                                                                   \
    ctx_free(&ctx);                                                                   \
    return ORC_OK
int orc_shade_hit(const orc_world *w, const orc_object *obj, const orc_comps *c, uint32_t remaining, double *out3)
{
    WORLD_CALL(shade_hit(w, &r, remaining, &ctx));
}
int orc_reflected_color(const orc_world *w, const orc_object *obj, const orc_comps *c, uint32_t remaining, double *out3)
{
    WORLD_CALL(reflected_color(w, &r, remaining, &ctx));
}
int orc_refracted_color(const orc_world *w, const orc_object *obj, const orc_comps *c, uint32_t remaining, double *out3)
{
    WORLD_CALL(refracted_color(w, &r, remaining, &ctx));
}

/* camera.rs:18-37 */
orc_camera *orc_camera_new(uint64_t hsize, uint64_t vsize, double fov)
{
    orc_camera *c = (orc_camera *)calloc(1, sizeof *c);
    c->hsize = hsize;
    c->vsize = vsize;
    c->field_of_view = fov;
    c->transform = m4_identity();
    c->inverse = m4_identity();
    c->from = point(0.0, 0.0, 0.0);
    c->to = point(0.0, 0.0, -1.0);
    c->up = vector(0.0, 1.0, 0.0);
    camera_initialize(c);
    jmp_buf jb;
    jmp_buf *saved = tl_jmp;
    if (setjmp(jb)) { tl_jmp = saved; free(c); return NULL; }
    tl_jmp = &jb;
    camera_update_transformations(c);
    tl_jmp = saved;
    return c;
}
/* camera.rs:39-55: the ratio and the product are f32 */
orc_camera *orc_camera_new_preview(const orc_camera *src)
{
    float ratio = (float)src->hsize / (float)src->vsize;
    orc_camera *c = (orc_camera *)calloc(1, sizeof *c);
    *c = *src;
    c->hsize = (uint64_t)cast_usize((double)(400.0f * ratio));
    c->vsize = 400;
    camera_initialize(c);
    jmp_buf jb;
    jmp_buf *saved = tl_jmp;
    if (setjmp(jb)) { tl_jmp = saved; free(c); return NULL; }
    tl_jmp = &jb;
    camera_update_transformations(c);
    tl_jmp = saved;
    return c;
}
void orc_camera_free(orc_camera *c) { free(c); }
int orc_camera_set_transform(orc_camera *c, const double *m16)
{
    API_ENTER();
    camera_set_transform(c, m4_load(m16));
    API_LEAVE();
    return ORC_OK;
}
/* camera.rs:123-136 */
int orc_camera_set_view(orc_camera *c, const double *f, const double *t, const double *u)
{
    API_ENTER();
    c->from = point(f[0], f[1], f[2]);
    c->to = point(t[0], t[1], t[2]);
    c->up = vector(u[0], u[1], u[2]);
    camera_update_transformations(c);
    API_LEAVE();
    return ORC_OK;
}
void orc_camera_get(const orc_camera *c, uint64_t *hsize, uint64_t *vsize, double *pixel_size, double *half_width,
                    double *half_height, double *inverse16)
{
    if (hsize) *hsize = c->hsize;
    if (vsize) *vsize = c->vsize;
    if (pixel_size) *pixel_size = c->pixel_size;
    if (half_width) *half_width = c->half_width;
    if (half_height) *half_height = c->half_height;
    if (inverse16) m4_store(c->inverse, inverse16);
}
void orc_camera_ray_for_pixel(const orc_camera *c, uint64_t x, uint64_t y, double *out)
{
    ray_t r = ray_for_pixel(c, x, y);
    out[0] = r.origin.x; out[1] = r.origin.y; out[2] = r.origin.z;
    out[3] = r.direction.x; out[4] = r.direction.y; out[5] = r.direction.z;
}

/* camera.rs:160-170 */
int orc_render_rows(const orc_camera *c, const orc_world *w, uint32_t depth, uint64_t row_begin, uint64_t row_end,
                    double *out_rgb, uint64_t *counters2)
{
    trace_ctx ctx;
    memset(&ctx, 0, sizeof ctx);
    jmp_buf jb;
    jmp_buf *saved = tl_jmp;
    if (setjmp(jb)) { tl_jmp = saved; ctx_free(&ctx); return ORC_PANIC; }
    tl_jmp = &jb;
    if (row_end > c->vsize) row_end = c->vsize;
    for (uint64_t y = row_begin; y < row_end; y++)
        for (uint64_t x = 0; x < c->hsize; x++) {
            color_t col = color_at(w, ray_for_pixel(c, x, y), depth, &ctx);
            double *px = out_rgb + 3 * (y * c->hsize + x); /* canvas.rs:32-34 */
            px[0] = col.r; px[1] = col.g; px[2] = col.b;
        }
    if (counters2) { counters2[0] += ctx.n_color; counters2[1] += ctx.n_shadow; }
    tl_jmp = saved;
    ctx_free(&ctx);
    return ORC_OK;
}

/* camera.rs:172-201 */
#define BAND_SIZE 10
typedef struct {
    const orc_camera *c;
    const orc_world *w;
    uint32_t depth;
    double *out;
    uint64_t n_bands;
    uint64_t next_band; /* atomic */
    uint64_t n_color, n_shadow;
    int status;
    pthread_mutex_t mu;
} band_job;

static void *band_worker(void *arg)
{
    band_job *job = (band_job *)arg;
    uint64_t counters[2] = { 0, 0 };
    int status = ORC_OK;
    for (;;) {
        uint64_t band = __atomic_fetch_add(&job->next_band, 1, __ATOMIC_RELAXED);
        if (band >= job->n_bands) break;
        if (orc_render_rows(job->c, job->w, job->depth, band * BAND_SIZE, (band + 1) * BAND_SIZE, job->out, counters) != ORC_OK) {
            status = ORC_PANIC;
            break;
        }
    }
    pthread_mutex_lock(&job->mu);
    job->n_color += counters[0];
    job->n_shadow += counters[1];
    if (status != ORC_OK) job->status = status;
    pthread_mutex_unlock(&job->mu);
    return NULL;
}
int orc_parallel_render(const orc_camera *c, const orc_world *w, uint32_t depth, int n_threads, double *out_rgb,
                        uint64_t *counters2)
{
    if (n_threads < 1) n_threads = 1;
    band_job job;
    memset(&job, 0, sizeof job);
    job.c = c; job.w = w; job.depth = depth; job.out = out_rgb;
    job.n_bands = (c->vsize + BAND_SIZE - 1) / BAND_SIZE;
    pthread_mutex_init(&job.mu, NULL);
    pthread_t *th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
    for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, band_worker, &job);
    for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    free(th);
    pthread_mutex_destroy(&job.mu);
    if (counters2) { counters2[0] += job.n_color; counters2[1] += job.n_shadow; }
    if (job.status != ORC_OK) snprintf(tl_panic, sizeof tl_panic, "a render worker panicked");
    return job.status;
}

/* canvas.rs:84-86 */
void orc_quantize_u8(const double *rgb, uint64_t n, uint8_t *out)
{
    for (uint64_t i = 0; i < n; i++) {
        double v = rgb[i] * 255.999;
        if (v != v || v <= 0.0) out[i] = 0;
        else if (v >= 255.0) out[i] = 255;
        else out[i] = (uint8_t)v;
    }
}
/* canvas.rs:52-71 */
void orc_quantize_ppm(const double *rgb, uint64_t n, int32_t *out)
{
    for (uint64_t i = 0; i < n; i++) {
        long long r = cast_isize(rgb[i] * 255.0);
        if (r > 255) r = 255;
        else if (r < 0) r = 0;
        out[i] = (int32_t)r;
    }
}
