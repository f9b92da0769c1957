/*
 * rt_oracle.h -- CPU restatement of the per-pixel render loop of malloc-man/ray-tracer.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker and the CPU
 * baseline.  Nothing in the product library links, loads or calls it; only the
 * test-suite, the smoke check and the benchmark's cpu_baseline leg do.
 *
 * Parity status: pinned to the reference's own known-answer unit tests (5-decimal
 * constants, tolerance 1e-5, plus the exact t-values those tests assert).  The
 * reference is a Rust crate and no Rust toolchain exists where this is built, so
 * agreement with the Rust binary beyond those constants is NOT pinned.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference checkout).  All arithmetic is IEEE f64, evaluated in the reference's
 * operation order with separate multiply and add; build with
 *   -O2 -ffp-contract=off -fno-fast-math
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OK 0
#define ORC_PANIC 1 /* the reference would have panicked; see orc_last_panic() */

/* shapes/objects.rs:14-20 */
enum { ORC_SPHERE = 0, ORC_PLANE = 1, ORC_CUBE = 2, ORC_CYLINDER = 3, ORC_CONE = 4 };
/* surfaces/patterns.rs:13-20 */
enum { ORC_SOLID = 0, ORC_STRIPE = 1, ORC_GRADIENT = 2, ORC_RING = 3, ORC_CHECKER3D = 4, ORC_TEST = 5 };
/* scalar material fields, surfaces/materials.rs:4-15 */
enum {
    ORC_AMBIENT = 0, ORC_DIFFUSE = 1, ORC_SPECULAR = 2, ORC_SHININESS = 3,
    ORC_REFLECTIVE = 4, ORC_TRANSPARENCY = 5, ORC_REFRACTIVE_INDEX = 6
};

typedef struct orc_object orc_object;
typedef struct orc_world orc_world;
typedef struct orc_group orc_group;
typedef struct orc_camera orc_camera;

const char *orc_last_panic(void);

/* ---- matrices (matrices/matrix4.rs, transformations.rs); 4x4 row-major double[16] ---- */
void orc_identity(double *out);
void orc_translation(double x, double y, double z, double *out);
void orc_scaling(double x, double y, double z, double *out);
void orc_rotation_x(double rad, double *out);
void orc_rotation_y(double rad, double *out);
void orc_rotation_z(double rad, double *out);
void orc_shearing(double xy, double xz, double yx, double yz, double zx, double zy, double *out);
void orc_mat4_mul(const double *a, const double *b, double *out);
void orc_mat4_transpose(const double *a, double *out);
double orc_mat4_determinant(const double *a);
int orc_mat4_invert(const double *a, double *out);
/* tuple = (x, y, z, v) with v = 1 point / 0 vector */
void orc_mat4_mul_tuple(const double *m, const double *tup4, double *out4);
int orc_view_transform(const double *from3, const double *to3, const double *up3, double *out);
int orc_normalize(const double *v3, double *out3);
void orc_reflect_vector(const double *v3, const double *n3, double *out3);

/* ---- objects (shapes/objects.rs, traits.rs, spheres.rs ... cones.rs) ---- */
orc_object *orc_object_new(int shape, double ymin, double ymax, int closed);
orc_object *orc_glass_sphere(void);
orc_object *orc_object_clone(const orc_object *o);
void orc_object_free(orc_object *o);
int orc_object_set_transform(orc_object *o, const double *m16);
int orc_object_set_transformation_list(orc_object *o, int index, double x);
/* the named helpers translate_x .. shear_zy (traits.rs:46-98); slot = list index */
int orc_object_transform_op(orc_object *o, int slot, double x);
void orc_object_set_color(orc_object *o, double r, double g, double b);
void orc_object_set_scalar(orc_object *o, int field, double value);
void orc_object_set_casts_shadow(orc_object *o, int casts);
void orc_object_set_pattern(orc_object *o, int type, const double *c1, const double *c2);
int orc_object_set_pattern_transform(orc_object *o, const double *m16);
void orc_object_get_matrices(const orc_object *o, double *transform, double *inverse, double *inverse_t);
void orc_object_get_pattern_inverse(const orc_object *o, double *inverse);
int orc_object_equal(const orc_object *a, const orc_object *b);
int orc_object_normal_at(const orc_object *o, const double *p3, double *out3);
/* world-space ray (ox,oy,oz,dx,dy,dz); returns number of t values written (<= 4) */
int orc_object_intersect(const orc_object *o, const double *ray6, double *out_t);
int orc_object_local_intersect(const orc_object *o, const double *ray6, double *out_t);
void orc_object_pattern_at_object(const orc_object *o, const double *p3, double *out3);
/* Pattern::pattern_at on a pattern-space point, for the pattern carried by o */
void orc_object_pattern_at(const orc_object *o, const double *p3, double *out3);

/* ---- lights and lighting (scenes/lights.rs) ---- */
int orc_lighting(const orc_object *o, const double *light_pos3, const double *light_int3,
                 const double *point3, const double *eyev3, const double *normalv3,
                 int in_shadow, double *out3);

/* ---- world (scenes/world.rs) ---- */
orc_world *orc_world_new(void);
orc_world *orc_world_default(void);
void orc_world_free(orc_world *w);
int orc_world_add_object(orc_world *w, const orc_object *o); /* copies; returns index */
/* Group, src/shapes/groups.rs; ObjectHolder, src/shapes/objectholders.rs.  Adding copies. */
orc_group *orc_group_new(void);
void orc_group_free(orc_group *g);
int orc_group_add_object(orc_group *g, const orc_object *o);
int orc_group_add_group(orc_group *g, const orc_group *child);
int orc_group_n_elements(const orc_group *g);
size_t orc_object_sizeof(void); /* bytes of one orc_object, for callers that receive copies */
int orc_group_set_transform(orc_group *g, const double *m16);
int orc_group_set_transformation_list(orc_group *g, int index, double x);
int orc_group_transform_op(orc_group *g, int slot, double x);
void orc_group_get_matrices(const orc_group *g, double *transform, double *inverse, double *inverse_t);
/* Group::intersect as written (recursive, every level sorts); out_objects receives the copies
 * with merged transformation lists; returns the count, -1 on panic */
int orc_group_intersect(const orc_group *g, const double *ray6, double *out_t, orc_object *out_objects, int cap);
/* World::add_group (world.rs:82-84): returns the index of the group's first leaf in the
 * flattened object list */
int orc_world_add_group(orc_world *w, const orc_group *g);
int orc_world_object_chain(const orc_world *w, int index, double *inverses16, int cap);
const orc_object *orc_world_object_base(const orc_world *w, int index);
void orc_world_add_light(orc_world *w, const double *pos3, const double *intensity3);
void orc_world_clear_lights(orc_world *w);
int orc_world_n_objects(const orc_world *w);
orc_object *orc_world_object(orc_world *w, int index); /* borrowed, mutable */
int orc_world_intersect(const orc_world *w, const double *ray6, double *out_t, int *out_obj, int cap);
int orc_world_color_at(const orc_world *w, const double *ray6, uint32_t remaining, double *out3);
int orc_world_is_shadowed(const orc_world *w, const double *p3, int *out);
/* batch of n rays; counters[0] += color_at calls, counters[1] += is_shadowed calls */
int orc_world_color_at_batch(const orc_world *w, const double *rays6, uint64_t n, uint32_t remaining,
                             double *out_rgb, uint64_t *counters2);

/* prepare_computations on a hand-made intersection list (world.rs:242-287) */
typedef struct orc_comps {
    double t, point[3], eyev[3], normalv[3], over_point[3], under_point[3], reflectv[3];
    double n1, n2;
    int32_t inside;
    int32_t object_index; /* index into the objs[] array given to orc_prepare_computations */
} orc_comps;
int orc_prepare_computations(int n, const double *ts, const orc_object *const *objs, int hit_index,
                             const double *ray6, orc_comps *out);
double orc_schlick(const orc_comps *c);
int orc_shade_hit(const orc_world *w, const orc_object *obj, const orc_comps *c, uint32_t remaining, double *out3);
int orc_reflected_color(const orc_world *w, const orc_object *obj, const orc_comps *c, uint32_t remaining, double *out3);
int orc_refracted_color(const orc_world *w, const orc_object *obj, const orc_comps *c, uint32_t remaining, double *out3);

/* ---- camera (scenes/camera.rs) ---- */
orc_camera *orc_camera_new(uint64_t hsize, uint64_t vsize, double fov);
orc_camera *orc_camera_new_preview(const orc_camera *c);
void orc_camera_free(orc_camera *c);
int orc_camera_set_transform(orc_camera *c, const double *m16);
int orc_camera_set_view(orc_camera *c, const double *from3, const double *to3, const double *up3);
void orc_camera_get(const orc_camera *c, uint64_t *hsize, uint64_t *vsize, double *pixel_size,
                    double *half_width, double *half_height, double *inverse16);
void orc_camera_ray_for_pixel(const orc_camera *c, uint64_t x, uint64_t y, double *out_ray6);

/* Camera::render (camera.rs:160-170) restricted to rows [row_begin,row_end); out is the
 * full-frame buffer, row-major, 3 doubles per pixel (canvas.rs:32-34). */
int orc_render_rows(const orc_camera *c, const orc_world *w, uint32_t depth, uint64_t row_begin,
                    uint64_t row_end, double *out_rgb, uint64_t *counters2);
/* Camera::parallel_render (camera.rs:172-201): 10-row bands handed out dynamically to
 * n_threads workers (a stand-in for Rayon's pool). */
int orc_parallel_render(const orc_camera *c, const orc_world *w, uint32_t depth, int n_threads,
                        double *out_rgb, uint64_t *counters2);

/* canvas.rs:84-86 / 98-101: (c * 255.999) as u8, saturating, NaN -> 0 */
void orc_quantize_u8(const double *rgb, uint64_t n_values, uint8_t *out);
/* canvas.rs:52-71: (c * 255.0) as isize then clamp to 0..255 (PPM rule) */
void orc_quantize_ppm(const double *rgb, uint64_t n_values, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif
