// rt_capi.cpp -- the flat C functions of include/rt_scene.h over the classes of rt.hpp.
// A Rust panic is an rt::Panic here; it is caught at this boundary and reported as status 1.
#include <atomic>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_scene.h"
#include "rt.hpp"

using namespace rt;

struct rt_object {
    Object o;
};
struct rt_world {
    World w;
};
struct rt_camera {
    Camera c;
};

namespace {
thread_local std::string tl_panic;

template <typename F> int guarded(F &&f)
{
    try {
        f();
        return 0;
    } catch (const std::exception &e) {
        tl_panic = e.what();
        return 1;
    }
}
Color col3(const double *c) { return color(c[0], c[1], c[2]); }
}  // namespace

extern "C" {

const char *rt_last_panic(void) { return tl_panic.c_str(); }

void rt_identity(double *out) { Matrix4::identity().store(out); }
void rt_translation(double x, double y, double z, double *out) { translation(x, y, z).store(out); }
void rt_scaling(double x, double y, double z, double *out) { scaling(x, y, z).store(out); }
void rt_rotation_x(double rad, double *out) { rotation_x(rad).store(out); }
void rt_rotation_y(double rad, double *out) { rotation_y(rad).store(out); }
void rt_rotation_z(double rad, double *out) { rotation_z(rad).store(out); }
void rt_shearing(double xy, double xz, double yx, double yz, double zx, double zy, double *out)
{
    shearing(xy, xz, yx, yz, zx, zy).store(out);
}
void rt_mat4_mul(const double *a, const double *b, double *out) { (Matrix4::convert(a) * Matrix4::convert(b)).store(out); }
void rt_mat4_transpose(const double *a, double *out) { Matrix4::convert(a).transpose().store(out); }
double rt_mat4_determinant(const double *a) { return Matrix4::convert(a).determinant(); }
int rt_mat4_invert(const double *a, double *out)
{
    return guarded([&] { Matrix4::convert(a).invert().store(out); });
}
void rt_mat4_mul_tuple(const double *m, const double *t, double *out)
{
    const Tuple r = Matrix4::convert(m) * Tuple{ t[0], t[1], t[2], (int)t[3] };
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = (double)r.v;
}
int rt_view_transform(const double *f, const double *t, const double *u, double *out)
{
    return guarded([&] { view_transform(point(f[0], f[1], f[2]), point(t[0], t[1], t[2]), vector(u[0], u[1], u[2])).store(out); });
}

rt_object *rt_object_new(int shape, double ymin, double ymax, int closed)
{
    if (shape < (int)ShapeKind::Sphere || shape > (int)ShapeKind::Cone) {
        tl_panic = "unknown shape " + std::to_string(shape);
        return nullptr;
    }
    Shape s;
    s.kind = (ShapeKind)shape;
    if (s.kind == ShapeKind::Cylinder || s.kind == ShapeKind::Cone) {
        s.min = ymin;
        s.max = ymax;
        s.closed = closed != 0;
    }
    return new rt_object{ Object(s) };
}
rt_object *rt_glass_sphere(void) { return new rt_object{ spheres::glass_sphere() }; }
rt_object *rt_object_clone(const rt_object *o) { return new rt_object{ o->o }; }
void rt_object_free(rt_object *o) { delete o; }
int rt_object_set_transform(rt_object *o, const double *m16)
{
    return guarded([&] { o->o.set_transform(Matrix4::convert(m16)); });
}
int rt_object_set_transformation_list(rt_object *o, int index, double x)
{
    return guarded([&] { o->o.set_transformation_list((size_t)index, x); });
}
int rt_object_transform_op(rt_object *o, int slot, double x)
{
    return guarded([&] {
        switch (slot) {
        case 0: o->o.translate_x(x); break;
        case 1: o->o.translate_y(x); break;
        case 2: o->o.translate_z(x); break;
        case 3: o->o.scale_x(x); break;
        case 4: o->o.scale_y(x); break;
        case 5: o->o.scale_z(x); break;
        case 6: o->o.rotate_x(x); break;
        case 7: o->o.rotate_y(x); break;
        case 8: o->o.rotate_z(x); break;
        case 9: o->o.shear_xy(x); break;
        case 10: o->o.shear_xz(x); break;
        case 11: o->o.shear_yx(x); break;
        case 12: o->o.shear_yz(x); break;
        case 13: o->o.shear_zx(x); break;
        case 14: o->o.shear_zy(x); break;
        default: throw Panic("no such transformation slot");
        }
    });
}
void rt_object_set_color(rt_object *o, double r, double g, double b) { o->o.set_color(color(r, g, b)); }
void rt_object_set_scalar(rt_object *o, int field, double value)
{
    switch (field) {
    case RT_AMBIENT: o->o.set_ambient(value); break;
    case RT_DIFFUSE: o->o.set_diffuse(value); break;
    case RT_SPECULAR: o->o.set_specular(value); break;
    case RT_SHININESS: o->o.set_shininess(value); break;
    case RT_REFLECTIVE: o->o.set_reflective(value); break;
    case RT_TRANSPARENCY: o->o.set_transparency(value); break;
    case RT_REFRACTIVE_INDEX: o->o.set_refractive_index(value); break;
    default: break;
    }
}
void rt_object_set_casts_shadow(rt_object *o, int casts) { o->o.set_casts_shadow(casts != 0); }
void rt_object_set_pattern(rt_object *o, int type, const double *c1, const double *c2)
{
    o->o.set_pattern(Pattern((PatternType)type, col3(c1), col3(c2)));
}
int rt_object_set_pattern_transform(rt_object *o, const double *m16)
{
    return guarded([&] { o->o.set_pattern_transform(Matrix4::convert(m16)); });
}
void rt_object_get_matrices(const rt_object *o, double *transform, double *inverse, double *inverse_t)
{
    if (transform) o->o.get_transform().store(transform);
    if (inverse) o->o.get_inverse_transform().store(inverse);
    if (inverse_t) o->o.get_inverse_transform_transposed().store(inverse_t);
}
void rt_object_get_pattern_inverse(const rt_object *o, double *inverse) { o->o.material.pattern.inverse_transform.store(inverse); }
int rt_object_equal(const rt_object *a, const rt_object *b) { return a->o == b->o ? 1 : 0; }

rt_world *rt_world_new(void) { return new rt_world; }
rt_world *rt_world_default(void)
{
    rt_world *w = new rt_world;
    w->w = World::new_default();
    return w;
}
void rt_world_free(rt_world *w)
{
    delete w;
}
int rt_world_add_object(rt_world *w, const rt_object *o)
{
    w->w.add_object(o->o);
    return (int)w->w.objects().size() - 1;
}
// ---- groups --------------------------------------------------------------------------------------
struct rt_group {
    Group g;
};
rt_group *rt_group_new(void) { return new rt_group; }
void rt_group_free(rt_group *g) { delete g; }
int rt_group_add_object(rt_group *g, const rt_object *o)
{
    g->g.add_object(o->o);
    return (int)g->g.elements().size() - 1;
}
int rt_group_add_group(rt_group *g, const rt_group *child)
{
    g->g.add_group(child->g);
    return (int)g->g.elements().size() - 1;
}
int rt_group_n_elements(const rt_group *g) { return (int)g->g.elements().size(); }
int rt_group_set_transform(rt_group *g, const double *m16)
{
    return guarded([&] { g->g.set_transform(Matrix4::convert(m16)); });
}
int rt_group_set_transformation_list(rt_group *g, int index, double x)
{
    return guarded([&] { g->g.set_transformation_list((size_t)index, x); });
}
int rt_group_transform_op(rt_group *g, int slot, double x)
{
    return guarded([&] {
        if (slot >= 3 && slot <= 5 && !(x != 0.0)) x = EPSILON;  // traits.rs:58-80
        g->g.set_transformation_list((size_t)slot, x);
    });
}
void rt_group_get_matrices(const rt_group *g, double *transform, double *inverse, double *inverse_t)
{
    if (transform) g->g.get_transform().store(transform);
    if (inverse) g->g.get_inverse_transform().store(inverse);
    if (inverse_t) g->g.get_inverse_transform_transposed().store(inverse_t);
}
int rt_world_add_group(rt_world *w, const rt_group *g)
{
    int first = -1;
    const int rc = guarded([&] { first = (int)w->w.add_group(g->g); });
    return rc ? -1 : first;
}
int rt_world_object_chain(const rt_world *w, int index, double *inverses16, int cap)
{
    if (index < 0 || index >= (int)w->w.objects().size()) return -1;
    const World::Chain &c = w->w.chain((size_t)index);
    if (c.inverses.empty()) return 0;
    const int depth = (int)c.inverses.size() - 1;
    for (int d = 0; d < depth && d < cap; ++d) c.inverses[(size_t)d].store(inverses16 + 16 * d);
    return depth;
}

void rt_world_add_light(rt_world *w, const double *p, const double *i)
{
    w->w.add_light(Light(point(p[0], p[1], p[2]), col3(i)));
}
void rt_world_clear_lights(rt_world *w) { w->w.lights().clear(); }
int rt_world_n_objects(const rt_world *w) { return (int)w->w.objects().size(); }

// A mutable view of an object stored in the world (the reference's tests mutate objects in
// place through a write lock, e.g. world.rs:404-420).  rt_object wraps exactly one Object, so
// the element itself serves as the handle; it stays valid until the next rt_world_add_object.
rt_object *rt_world_object(rt_world *w, int index)
{
    if (index < 0 || index >= (int)w->w.objects().size()) return nullptr;
    static_assert(sizeof(rt_object) == sizeof(Object), "rt_object wraps one Object");
    return reinterpret_cast<rt_object *>(&w->w.objects()[(size_t)index]);
}

int rt_world_color_at_batch(const rt_world *w, const double *rays6, uint64_t n, uint32_t remaining, int precision,
                            double *out_rgb, uint64_t *counters2)
{
    return guarded([&] {
        rtx_context *ctx = device_context();
        const FlatScene flat = w->w.flatten();
        const rtx_scene_desc d = flat.desc();
        rtx_scene *scene = nullptr;
        if (rtx_scene_upload(ctx, &d, &scene) != RTX_OK) throw Panic(rtx_last_error());
        rtx_stats st;
        const int rc = rtx_trace_rays(ctx, scene, rays6, n, remaining, precision, out_rgb, &st);
        rtx_scene_destroy(scene);
        if (rc != RTX_OK) throw Panic(rtx_last_error());
        if (counters2) {
            counters2[0] += st.color_rays;
            counters2[1] += st.shadow_rays;
        }
    });
}

int rt_world_probe(const rt_world *w, const double *rays6, uint64_t n, rtx_probe *out)
{
    return guarded([&] {
        rtx_context *ctx = device_context();
        const FlatScene flat = w->w.flatten();
        const rtx_scene_desc d = flat.desc();
        rtx_scene *scene = nullptr;
        if (rtx_scene_upload(ctx, &d, &scene) != RTX_OK) throw Panic(rtx_last_error());
        const int rc = rtx_probe_rays(ctx, scene, rays6, n, out);
        rtx_scene_destroy(scene);
        if (rc != RTX_OK) throw Panic(rtx_last_error());
    });
}

int64_t rt_world_flat(const rt_world *w, int field, double *out, int64_t cap)
{
    const FlatScene f = w->w.flatten();
    std::vector<double> v;
    const size_t n = f.shape_type.size();
    switch (field) {
    case RT_FLAT_INVERSE: v = f.inverse; break;
    case RT_FLAT_INVERSE_T: v = f.inverse_transposed; break;
    case RT_FLAT_PATTERN_INVERSE: v = f.pattern_inverse; break;
    case RT_FLAT_COLOR: v = f.color; break;
    case RT_FLAT_MATERIAL:
        for (size_t i = 0; i < n; ++i)
            v.insert(v.end(), { f.ambient[i], f.diffuse[i], f.specular[i], f.shininess[i], f.reflective[i], f.transparency[i],
                                f.refractive_index[i] });
        break;
    case RT_FLAT_YRANGE:
        for (size_t i = 0; i < n; ++i) v.insert(v.end(), { f.ymin[i], f.ymax[i] });
        break;
    case RT_FLAT_TAGS:
        for (size_t i = 0; i < n; ++i)
            v.insert(v.end(), { (double)f.shape_type[i], (double)f.closed[i], (double)f.casts_shadow[i], (double)f.pattern_type[i],
                                f.eq_class.empty() ? (double)i : (double)f.eq_class[i] });
        break;
    case RT_FLAT_LIGHTS:
        for (size_t i = 0; i < f.light_position.size() / 3; ++i)
            v.insert(v.end(), { f.light_position[i * 3], f.light_position[i * 3 + 1], f.light_position[i * 3 + 2],
                                f.light_intensity[i * 3], f.light_intensity[i * 3 + 1], f.light_intensity[i * 3 + 2] });
        break;
    case RT_FLAT_PATTERN_COLORS:
        for (size_t i = 0; i < n; ++i)
            v.insert(v.end(), { f.pattern_color1[i * 3], f.pattern_color1[i * 3 + 1], f.pattern_color1[i * 3 + 2],
                                f.pattern_color2[i * 3], f.pattern_color2[i * 3 + 1], f.pattern_color2[i * 3 + 2] });
        break;
    case RT_FLAT_CHAIN:
        for (size_t i = 0; i < n; ++i)
            v.insert(v.end(), { f.chain_depth.empty() ? 0.0 : (double)f.chain_depth[i], f.chain_first.empty() ? 0.0 : (double)f.chain_first[i] });
        break;
    case RT_FLAT_CHAIN_INVERSE: v = f.chain_inverse; break;
    default: return -1;
    }
    if (out)
        for (int64_t i = 0; i < cap && i < (int64_t)v.size(); ++i) out[i] = v[(size_t)i];
    return (int64_t)v.size();
}

rt_camera *rt_camera_new(uint64_t hsize, uint64_t vsize, double fov)
{
    rt_camera *c = nullptr;
    guarded([&] { c = new rt_camera{ Camera((size_t)hsize, (size_t)vsize, fov) }; });
    return c;
}
rt_camera *rt_camera_new_preview(const rt_camera *src)
{
    rt_camera *c = nullptr;
    guarded([&] { c = new rt_camera{ Camera::new_preview(src->c) }; });
    return c;
}
void rt_camera_free(rt_camera *c) { delete c; }
int rt_camera_set_transform(rt_camera *c, const double *m16)
{
    return guarded([&] { c->c.set_transform(Matrix4::convert(m16)); });
}
int rt_camera_set_view(rt_camera *c, const double *f, const double *t, const double *u)
{
    return guarded([&] { c->c.set_view(point(f[0], f[1], f[2]), point(t[0], t[1], t[2]), vector(u[0], u[1], u[2])); });
}
void rt_camera_get(const rt_camera *c, uint64_t *hsize, uint64_t *vsize, double *pixel_size, double *half_width,
                   double *half_height, double *inverse16)
{
    if (hsize) *hsize = c->c.get_hsize();
    if (vsize) *vsize = c->c.get_vsize();
    if (pixel_size) *pixel_size = c->c.pixel_size();
    if (half_width) *half_width = c->c.half_width();
    if (half_height) *half_height = c->c.half_height();
    if (inverse16) c->c.get_inverse_transform().store(inverse16);
}
int rt_camera_ray_for_pixel(const rt_camera *c, uint64_t x, uint64_t y, double *out)
{
    return guarded([&] {
        const Ray r = c->c.ray_for_pixel((size_t)x, (size_t)y);
        out[0] = r.origin.x; out[1] = r.origin.y; out[2] = r.origin.z;
        out[3] = r.direction.x; out[4] = r.direction.y; out[5] = r.direction.z;
    });
}
int rt_camera_render_rows(const rt_camera *c, const rt_world *w, uint32_t depth, int precision, uint64_t row_begin,
                          uint64_t row_end, double *out_rgb, rtx_stats *stats)
{
    return guarded([&] {
        if (precision == RTX_F64) {
            c->c.render_rows(w->w, depth, (uint32_t)row_begin, (uint32_t)row_end, out_rgb, stats);
            return;
        }
        rtx_context *ctx = device_context();
        const FlatScene flat = w->w.flatten();
        const rtx_scene_desc sd = flat.desc();
        rtx_scene *scene = nullptr;
        if (rtx_scene_upload(ctx, &sd, &scene) != RTX_OK) throw Panic(rtx_last_error());
        const rtx_camera_desc cd = c->c.desc();
        const size_t n = (size_t)(row_end - row_begin) * cd.hsize * 3;
        std::vector<float> tmp(n);
        const int rc = rtx_render(ctx, scene, &cd, depth, RTX_F32, (uint32_t)row_begin, (uint32_t)row_end, tmp.data(),
                                  n * sizeof(float), nullptr, nullptr, stats);
        rtx_scene_destroy(scene);
        if (rc != RTX_OK) throw Panic(rtx_last_error());
        for (size_t i = 0; i < n; ++i) out_rgb[i] = (double)tmp[i];
    });
}
int rt_camera_parallel_render(const rt_camera *c, const rt_world *w, double *out_rgb, uint64_t *tracker, rtx_stats *stats)
{
    return guarded([&] {
        std::atomic<size_t> counter{ 0 };
        const Canvas img = c->c.parallel_render(w->w, tracker ? &counter : nullptr, stats, RTX_F64);
        std::memcpy(out_rgb, img.pixels.data(), img.pixels.size() * sizeof(Color));
        if (tracker) *tracker = counter.load();
    });
}

}  // extern "C"
