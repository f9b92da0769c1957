// rt.hpp -- host-side mirror of the reference crate's public scene API (its `prelude`,
// src/main.rs:9-77): tuples, matrices, colours, patterns, materials, objects, lights, World
// and Camera, with the names, argument meaning and panic conditions of the Rust items.  It
// exists because the Rust crate cannot be compiled where this library is built: it plays the
// crate's role above the C ABI of include/rtx.h.  Scene construction (matrix composition,
// cofactor inverse) runs here on the host in the reference's operation order; every render
// entry point flattens the World and hands it to the GPU through that ABI.  A Rust `panic!`
// becomes rt::Panic.
#pragma once

#include <array>
#include <memory>
#include <atomic>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtx.h"

namespace rt {

constexpr double EPSILON = 0.00001;              // src/utils.rs:1
constexpr uint32_t DEFAULT_RECURSION_DEPTH = 5;  // src/scenes/world.rs:3

struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

bool approx_eq(double a, double b);  // src/utils.rs:7-11

// ---- src/matrices/tuples.rs -------------------------------------------------------------
struct Tuple {
    double x = 0, y = 0, z = 0;
    int v = 0;  // 1 = point, 0 = vector
    bool is_vector() const { return v == 0; }
    Tuple vectorize() const { return { x, y, z, 0 }; }
    double magnitude() const;
    Tuple normalize() const;
    Tuple xprod(const Tuple &o) const;
    Tuple reflect_vector(const Tuple &normal) const;
    bool operator==(const Tuple &o) const;
};
Tuple tuple(double x, double y, double z, int v);
Tuple point(double x, double y, double z);
Tuple vector(double x, double y, double z);
Tuple origin();
Tuple operator+(const Tuple &a, const Tuple &b);
Tuple operator-(const Tuple &a, const Tuple &b);
Tuple operator-(const Tuple &a);
Tuple operator*(const Tuple &a, double s);
double operator*(const Tuple &a, const Tuple &b);  // dot product
Tuple operator/(const Tuple &a, double s);

// ---- src/surfaces/colors.rs ---------------------------------------------------------------
struct Color {
    double red = 0, green = 0, blue = 0;
    bool operator==(const Color &o) const;
};
Color color(double r, double g, double b);
Color black();
Color white();
Color operator+(const Color &a, const Color &b);
Color operator-(const Color &a, const Color &b);
Color operator*(const Color &a, const Color &b);
Color operator*(const Color &a, double s);

// ---- src/matrices/matrix4.rs, transformations.rs ----------------------------------------
struct Matrix4 {
    std::array<std::array<double, 4>, 4> vals{};
    static Matrix4 identity();
    static Matrix4 convert(const double *row_major16);
    double val_at(size_t row, size_t col) const;
    void write(double val, size_t row, size_t col);
    Matrix4 transpose() const;
    double determinant() const;
    bool is_invertible() const;
    Matrix4 invert() const;  // panics when |det| < EPSILON
    Matrix4 translate(double x, double y, double z) const;
    Matrix4 scale(double x, double y, double z) const;
    Matrix4 rotate_x(double rad) const;
    Matrix4 rotate_y(double rad) const;
    Matrix4 rotate_z(double rad) const;
    bool operator==(const Matrix4 &o) const;
    void store(double *row_major16) const;
};
Matrix4 operator*(const Matrix4 &a, const Matrix4 &b);
Tuple operator*(const Matrix4 &a, const Tuple &t);
Matrix4 translation(double x, double y, double z);
Matrix4 scaling(double x, double y, double z);
Matrix4 rotation_x(double rad);
Matrix4 rotation_y(double rad);
Matrix4 rotation_z(double rad);
Matrix4 shearing(double xy, double xz, double yx, double yz, double zx, double zy);
Matrix4 view_transform(const Tuple &from, const Tuple &to, const Tuple &up);

// ---- src/rays.rs ----------------------------------------------------------------------------
struct Ray {
    Tuple origin, direction;
    Ray(const Tuple &o, const Tuple &d);
    Tuple position(double t) const;
    Ray transform(const Matrix4 &m) const;
};

// ---- src/surfaces/patterns.rs ---------------------------------------------------------------
enum class PatternType : uint8_t { Solid = 0, Stripe = 1, Gradient = 2, Ring = 3, Checker3d = 4, Test = 5 };
struct Pattern {
    PatternType pattern_type = PatternType::Solid;
    Matrix4 transform = Matrix4::identity();
    Matrix4 inverse_transform = Matrix4::identity();
    Color color1, color2;
    Pattern() = default;
    Pattern(PatternType t, const Color &a, const Color &b) : pattern_type(t), color1(a), color2(b) {}
    Pattern &set_transform(const Matrix4 &m);
    bool operator==(const Pattern &o) const;
};
Pattern solid();
Pattern stripe(const Color &a, const Color &b);
Pattern gradient(const Color &a, const Color &b);
Pattern ring(const Color &a, const Color &b);
Pattern checker_3d(const Color &a, const Color &b);
Pattern test_pattern();

// ---- src/surfaces/materials.rs --------------------------------------------------------------
struct Material {
    Color color = white();
    double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
    Pattern pattern = solid();
    double reflective = 0.0, transparency = 0.0, refractive_index = 1.0;
    bool casts_shadow = true;
    bool operator==(const Material &o) const;
};

// ---- src/shapes/objects.rs, traits.rs -------------------------------------------------------
enum class ShapeKind : uint8_t { Sphere = 0, Plane = 1, Cube = 2, Cylinder = 3, Cone = 4 };
struct Shape {
    ShapeKind kind = ShapeKind::Sphere;
    double min = 0, max = 0;  // Cylinder / Cone
    bool closed = false;
    bool operator==(const Shape &o) const;
};

class Object {
public:
    explicit Object(const Shape &shape);
    Shape shape;
    Material material;

    // Transformable (src/shapes/traits.rs)
    void set_transform(const Matrix4 &m);
    void set_transformation_list(size_t index, double x);
    void update_transform();
    void combine_transforms(const std::array<double, 15> &list);
    void translate_x(double v) { set_transformation_list(0, v); }
    void translate_y(double v) { set_transformation_list(1, v); }
    void translate_z(double v) { set_transformation_list(2, v); }
    void scale_x(double v) { set_transformation_list(3, v != 0.0 ? v : EPSILON); }
    void scale_y(double v) { set_transformation_list(4, v != 0.0 ? v : EPSILON); }
    void scale_z(double v) { set_transformation_list(5, v != 0.0 ? v : EPSILON); }
    void rotate_x(double v) { set_transformation_list(6, v); }
    void rotate_y(double v) { set_transformation_list(7, v); }
    void rotate_z(double v) { set_transformation_list(8, v); }
    void shear_xy(double v) { set_transformation_list(9, v); }
    void shear_xz(double v) { set_transformation_list(10, v); }
    void shear_yx(double v) { set_transformation_list(11, v); }
    void shear_yz(double v) { set_transformation_list(12, v); }
    void shear_zx(double v) { set_transformation_list(13, v); }
    void shear_zy(double v) { set_transformation_list(14, v); }
    const Matrix4 &get_transform() const { return transform_; }
    const Matrix4 &get_inverse_transform() const { return inverse_; }
    const Matrix4 &get_inverse_transform_transposed() const { return inverse_t_; }
    const std::array<double, 15> &transformation_list_all() const { return list_; }

    // material accessors (src/shapes/objects.rs:54-168)
    Object &set_material(const Material &m) { material = m; return *this; }
    Object &set_color(const Color &c) { material.color = c; return *this; }
    Object &set_ambient(double v) { material.ambient = v; return *this; }
    Object &set_diffuse(double v) { material.diffuse = v; return *this; }
    Object &set_specular(double v) { material.specular = v; return *this; }
    Object &set_shininess(double v) { material.shininess = v; return *this; }
    Object &set_pattern(const Pattern &p) { material.pattern = p; return *this; }
    Object &set_pattern_transform(const Matrix4 &m) { material.pattern.set_transform(m); return *this; }
    Object &set_reflective(double v) { material.reflective = v; return *this; }
    Object &set_transparency(double v) { material.transparency = v; return *this; }
    Object &set_refractive_index(double v) { material.refractive_index = v; return *this; }
    Object &set_casts_shadow(bool v) { material.casts_shadow = v; return *this; }

    bool operator==(const Object &o) const;  // derived PartialEq, objects.rs:3

private:
    Matrix4 transform_ = Matrix4::identity();
    Matrix4 inverse_ = Matrix4::identity();
    Matrix4 inverse_t_ = Matrix4::identity();
    std::array<double, 15> list_{ { 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0 } };
};

namespace spheres { Object new_(); Object glass_sphere(); }
namespace planes { Object new_(); }
namespace cubes { Object new_(); }
namespace cylinders { Object new_(double min, double max, bool closed); Object new_unbounded(); }
namespace cones { Object new_(double min, double max, bool closed); Object new_unbounded(); }

// ---- src/shapes/groups.rs, src/shapes/objectholders.rs ------------------------------------------
class Group;
// enum ObjectHolder { Object(Object), Group(Group) } (objectholders.rs:5-8)
struct ObjectHolder {
    bool is_group = false;
    Object object{ Shape{} };
    std::shared_ptr<const Group> group;  // value semantics: a holder never shares a mutable group
};
class Group {
public:
    Group() = default;  // Group::new_empty, groups.rs:30-43
    void add_object(const Object &o);           // groups.rs:49-51
    void add_group(const Group &g);             // groups.rs:53-55
    const std::vector<ObjectHolder> &elements() const { return group_; }

    // Transformable (src/shapes/traits.rs), as for Object
    void set_transform(const Matrix4 &m);
    void set_transformation_list(size_t index, double x);
    void update_transform();
    void translate_x(double v) { set_transformation_list(0, v); }
    void translate_y(double v) { set_transformation_list(1, v); }
    void translate_z(double v) { set_transformation_list(2, v); }
    void scale_x(double v) { set_transformation_list(3, v != 0.0 ? v : EPSILON); }
    void scale_y(double v) { set_transformation_list(4, v != 0.0 ? v : EPSILON); }
    void scale_z(double v) { set_transformation_list(5, v != 0.0 ? v : EPSILON); }
    void rotate_x(double v) { set_transformation_list(6, v); }
    void rotate_y(double v) { set_transformation_list(7, v); }
    void rotate_z(double v) { set_transformation_list(8, v); }
    const Matrix4 &get_transform() const { return transform_; }
    const Matrix4 &get_inverse_transform() const { return inverse_; }
    const Matrix4 &get_inverse_transform_transposed() const { return inverse_t_; }
    const std::array<double, 15> &transformation_list_all() const { return list_; }

private:
    std::vector<ObjectHolder> group_;
    Matrix4 transform_ = Matrix4::identity();
    Matrix4 inverse_ = Matrix4::identity();
    Matrix4 inverse_t_ = Matrix4::identity();
    std::array<double, 15> list_{ { 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0 } };
};

// ---- src/scenes/lights.rs -------------------------------------------------------------------
struct Light {
    Tuple position;
    Color intensity;
    Light(const Tuple &p, const Color &i) : position(p), intensity(i) {}
};

// ---- src/scenes/canvas.rs -------------------------------------------------------------------
struct Canvas {
    size_t width = 0, height = 0;
    std::vector<Color> pixels;  // row-major, index y * width + x
    Canvas(size_t w, size_t h) : width(w), height(h), pixels(w * h) {}
    size_t xy_to_1d(size_t x, size_t y) const { return y * width + x; }
    void write_pixel(size_t x, size_t y, const Color &c) { pixels.at(xy_to_1d(x, y)) = c; }
    Color pixel_at(size_t x, size_t y) const { return pixels.at(xy_to_1d(x, y)); }
};

// ---- the GPU behind the API -------------------------------------------------------------------
// One rtx_context per process and device, created on first use.
rtx_context *device_context(int device_id = -1);

// The flattened World handed to rtx_scene_upload: owning arrays plus the view over them.
struct FlatScene {
    std::vector<uint8_t> shape_type, closed, casts_shadow, pattern_type;
    std::vector<double> inverse, inverse_transposed, ymin, ymax, color, ambient, diffuse, specular, shininess,
        reflective, transparency, refractive_index, pattern_inverse, pattern_color1, pattern_color2, light_position,
        light_intensity;
    std::vector<uint32_t> eq_class;
    // objects inside groups (empty when the world has none): see rtx_scene_desc
    std::vector<uint8_t> chain_depth;
    std::vector<uint32_t> chain_first;
    std::vector<double> chain_inverse;
    rtx_scene_desc desc() const;
};

// ---- src/scenes/world.rs ----------------------------------------------------------------------
class World {
public:
    World() = default;
    World(std::vector<Object> objects, std::vector<Light> lights)
        : objects_(std::move(objects)), chains_(objects_.size()), lights_(std::move(lights)) {}
    static World new_default();
    void add_object(const Object &o) { objects_.push_back(o); chains_.emplace_back(); }
    // World::add_group (world.rs:82-84).  The group's objects are appended to objects() depth
    // first -- the order the nested stable sorts of Group::local_intersect (groups.rs:73-84) and
    // intersect_world (world.rs:98-107) leave equal t in -- as the copies the reference returns in
    // its Intersections: the child with every enclosing group's transformation list merged in
    // (combine_transforms, traits.rs:35-44), innermost group first.  How the ray reaches the child
    // is kept beside it (chain()).  Returns the index of the first such object.
    size_t add_group(const Group &g);
    struct Chain {
        std::vector<Matrix4> inverses;  // enclosing groups, outermost first, then the child's own; empty for a plain object
    };
    const Chain &chain(size_t index) const { return chains_.at(index); }
    void add_light(const Light &l) { lights_.push_back(l); }
    std::vector<Object> &objects() { return objects_; }
    const std::vector<Object> &objects() const { return objects_; }
    std::vector<Light> &lights() { return lights_; }
    const std::vector<Light> &lights() const { return lights_; }

    FlatScene flatten() const;
    // World::color_at for a batch of rays, on the GPU (world.rs:136-144)
    std::vector<Color> color_at(const std::vector<Ray> &rays, uint32_t remaining, int precision = RTX_F64,
                                rtx_stats *stats = nullptr) const;
    Color color_at(const Ray &ray, uint32_t remaining) const;

private:
    void add_leaves(const Group &g, std::vector<Matrix4> &inverses, std::vector<const std::array<double, 15> *> &lists);
    std::vector<Object> objects_;
    std::vector<Chain> chains_;  // parallel to objects_
    std::vector<Light> lights_;
};

// ---- src/scenes/camera.rs ---------------------------------------------------------------------
class Camera {
public:
    Camera(size_t hsize, size_t vsize, double field_of_view);
    static Camera new_preview(const Camera &c);
    size_t get_hsize() const { return hsize_; }
    size_t get_vsize() const { return vsize_; }
    double get_fov() const { return fov_; }
    Camera &set_hsize(size_t h) { hsize_ = h; initialize(); return *this; }
    Camera &set_vsize(size_t v) { vsize_ = v; initialize(); return *this; }
    Camera &set_fov(double f) { fov_ = f; initialize(); return *this; }
    void set_transform(const Matrix4 &m);
    const Matrix4 &get_transform() const { return transform_; }
    const Matrix4 &get_inverse_transform() const { return inverse_; }
    Tuple get_from() const { return components_[0]; }
    Tuple get_to() const { return components_[1]; }
    Tuple get_up() const { return components_[2]; }
    void set_from(const Tuple &t) { components_[0] = t; update_transformations(); }
    void set_to(const Tuple &t) { components_[1] = t; update_transformations(); }
    void set_up(const Tuple &t) { components_[2] = t; update_transformations(); }
    // set_from + set_to + set_up with a single update_transformations at the end
    void set_view(const Tuple &from, const Tuple &to, const Tuple &up)
    {
        components_ = { { from, to, up } };
        update_transformations();
    }
    double pixel_size() const { return pixel_size_; }
    double half_width() const { return half_width_; }
    double half_height() const { return half_height_; }
    rtx_camera_desc desc() const;

    // camera.rs:147-158, evaluated by the same device code the render kernel uses
    Ray ray_for_pixel(size_t x, size_t y) const;
    // camera.rs:160-170
    Canvas render(const World &world) const;
    // camera.rs:172-201; tracker counts finished pixels
    Canvas parallel_render(const World &world, std::atomic<size_t> *tracker = nullptr, rtx_stats *stats = nullptr,
                           int precision = RTX_F64) const;
    // camera.rs:203-222 (recursion depth 2)
    Canvas preview_parallel_render(const World &world) const;
    // rows [row_begin, row_end) into a caller buffer of (row_end - row_begin) * hsize * 3 doubles
    void render_rows(const World &world, uint32_t depth, uint32_t row_begin, uint32_t row_end, double *out,
                     rtx_stats *stats = nullptr) const;

private:
    void initialize();
    void update_transformations();
    size_t hsize_, vsize_;
    double fov_;
    Matrix4 transform_ = Matrix4::identity(), inverse_ = Matrix4::identity();
    std::array<Tuple, 3> components_;
    double pixel_size_ = 0, half_width_ = 0, half_height_ = 0;
};

}  // namespace rt
