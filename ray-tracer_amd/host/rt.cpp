// rt.cpp -- implementation of the host-side mirror declared in rt.hpp.
// Scene-construction arithmetic keeps the reference's operation order (no FMA: build with
// -ffp-contract=off) so that the matrices handed to the GPU are the ones the Rust crate
// would have cached.
#include "rt.hpp"

#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

namespace rt {

bool approx_eq(double a, double b) { return std::fabs(a - b) < EPSILON; }

// ---- tuples -----------------------------------------------------------------------------
// src/matrices/tuples.rs:12-42
Tuple tuple(double x, double y, double z, int v)
{
    if (v != 0 && v != 1) throw Panic("Attempted to create tuple with v: " + std::to_string(v) + ". Must be 0 or 1");
    return { x, y, z, v };
}
Tuple point(double x, double y, double z) { return { x, y, z, 1 }; }
Tuple vector(double x, double y, double z) { return { x, y, z, 0 }; }
Tuple origin() { return point(0.0, 0.0, 0.0); }
// tuples.rs:58-64
double Tuple::magnitude() const
{
    if (!is_vector()) throw Panic("Cannot get magnitude of point");
    return std::sqrt(x * x + y * y + z * z + (double)(v * v));
}
// tuples.rs:66-73
Tuple Tuple::normalize() const
{
    if (!is_vector()) throw Panic("Cannot normalize a point");
    const double mag = magnitude();
    return vector(x / mag, y / mag, z / mag);
}
// tuples.rs:76-85
Tuple Tuple::xprod(const Tuple &o) const
{
    if (!is_vector() || !o.is_vector()) throw Panic("Need two vectors to compute cross product");
    return vector(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x);
}
// tuples.rs:87-91
Tuple Tuple::reflect_vector(const Tuple &normal) const
{
    const double d = 2.0 * (*this * normal);
    return *this - normal * d;
}
// tuples.rs:94-104
bool Tuple::operator==(const Tuple &o) const { return approx_eq(x, o.x) && approx_eq(y, o.y) && approx_eq(z, o.z) && v == o.v; }
Tuple operator+(const Tuple &a, const Tuple &b) { return tuple(a.x + b.x, a.y + b.y, a.z + b.z, a.v + b.v); }
Tuple operator-(const Tuple &a, const Tuple &b) { return tuple(a.x - b.x, a.y - b.y, a.z - b.z, a.v - b.v); }
Tuple operator-(const Tuple &a)
{
    if (!a.is_vector()) throw Panic("Cannot negate a point");
    return tuple(-a.x, -a.y, -a.z, -a.v);
}
Tuple operator*(const Tuple &a, double s)
{
    if (!a.is_vector()) throw Panic("Cannot multiply a point");
    return vector(a.x * s, a.y * s, a.z * s);
}
double operator*(const Tuple &a, const Tuple &b)
{
    if (!a.is_vector() || !b.is_vector()) throw Panic("Need two vectors to compute dot product");
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
Tuple operator/(const Tuple &a, double s)
{
    if (!a.is_vector()) throw Panic("Cannot divide a point");
    return vector(a.x / s, a.y / s, a.z / s);
}

// ---- colours ----------------------------------------------------------------------------
Color color(double r, double g, double b) { return { r, g, b }; }
Color black() { return { 0.0, 0.0, 0.0 }; }
Color white() { return { 1.0, 1.0, 1.0 }; }
Color operator+(const Color &a, const Color &b) { return { a.red + b.red, a.green + b.green, a.blue + b.blue }; }
Color operator-(const Color &a, const Color &b) { return { a.red - b.red, a.green - b.green, a.blue - b.blue }; }
Color operator*(const Color &a, const Color &b) { return { a.red * b.red, a.green * b.green, a.blue * b.blue }; }
Color operator*(const Color &a, double s) { return { a.red * s, a.green * s, a.blue * s }; }
bool Color::operator==(const Color &o) const { return approx_eq(red, o.red) && approx_eq(green, o.green) && approx_eq(blue, o.blue); }

// ---- matrices -----------------------------------------------------------------------------
namespace {

// Determinant by cofactor expansion along the first column, accumulated from 0.0, with the
// cofactor's sign applied to the minor before the multiply
// (src/matrices/matrix2.rs:23-25, matrix3.rs:47-67, matrix4.rs:78-98).
template <size_t N> struct Square {
    double a[N][N];
};
template <size_t N> double det(const Square<N> &m);
template <size_t N> Square<N - 1> drop(const Square<N> &m, size_t row, size_t col)
{
    Square<N - 1> s;
    size_t r = 0;
    for (size_t i = 0; i < N; ++i) {
        if (i == row) continue;
        size_t c = 0;
        for (size_t j = 0; j < N; ++j) {
            if (j == col) continue;
            s.a[r][c++] = m.a[i][j];
        }
        ++r;
    }
    return s;
}
template <size_t N> double cofactor(const Square<N> &m, size_t row, size_t col)
{
    const double minor = det(drop(m, row, col));
    return ((row + col) % 2 == 1) ? -minor : minor;
}
template <> double det<2>(const Square<2> &m) { return m.a[0][0] * m.a[1][1] - m.a[1][0] * m.a[0][1]; }
template <size_t N> double det(const Square<N> &m)
{
    double d = 0.0;
    for (size_t i = 0; i < N; ++i) d += m.a[i][0] * cofactor(m, i, 0);
    return d;
}
Square<4> to_square(const Matrix4 &m)
{
    Square<4> s;
    for (size_t i = 0; i < 4; ++i)
        for (size_t j = 0; j < 4; ++j) s.a[i][j] = m.vals[i][j];
    return s;
}

}  // namespace

Matrix4 Matrix4::identity()
{
    Matrix4 m;
    for (size_t i = 0; i < 4; ++i) m.vals[i][i] = 1.0;
    return m;
}
Matrix4 Matrix4::convert(const double *p)
{
    Matrix4 m;
    for (size_t i = 0; i < 4; ++i)
        for (size_t j = 0; j < 4; ++j) m.vals[i][j] = p[i * 4 + j];
    return m;
}
void Matrix4::store(double *p) const
{
    for (size_t i = 0; i < 4; ++i)
        for (size_t j = 0; j < 4; ++j) p[i * 4 + j] = vals[i][j];
}
double Matrix4::val_at(size_t row, size_t col) const
{
    if (row >= 4 || col >= 4) throw Panic("Read index error. Size is 4x4.");
    return vals[row][col];
}
void Matrix4::write(double val, size_t row, size_t col)
{
    if (row >= 4 || col >= 4) throw Panic("Write index error. Size is 4x4.");
    vals[row][col] = val;
}
// matrix4.rs:44-52
Matrix4 Matrix4::transpose() const
{
    Matrix4 r;
    for (size_t i = 0; i < 4; ++i)
        for (size_t j = 0; j < 4; ++j) r.vals[j][i] = vals[i][j];
    return r;
}
double Matrix4::determinant() const { return det(to_square(*this)); }
bool Matrix4::is_invertible() const { return !approx_eq(determinant(), 0.0); }
// matrix4.rs:105-119
Matrix4 Matrix4::invert() const
{
    if (!is_invertible()) throw Panic("Attempted to invert a non-invertible matrix");
    const Square<4> s = to_square(*this);
    const double determinant = det(s);
    Matrix4 r;
    for (size_t i = 0; i < 4; ++i)
        for (size_t j = 0; j < 4; ++j) r.vals[j][i] = cofactor(s, i, j) / determinant;
    return r;
}
// matrix4.rs:121-139: the fluent helpers pre-multiply
Matrix4 Matrix4::translate(double x, double y, double z) const { return translation(x, y, z) * *this; }
Matrix4 Matrix4::scale(double x, double y, double z) const { return scaling(x, y, z) * *this; }
Matrix4 Matrix4::rotate_x(double rad) const { return rotation_x(rad) * *this; }
Matrix4 Matrix4::rotate_y(double rad) const { return rotation_y(rad) * *this; }
Matrix4 Matrix4::rotate_z(double rad) const { return rotation_z(rad) * *this; }
bool Matrix4::operator==(const Matrix4 &o) const
{
    for (size_t i = 0; i < 4; ++i)
        for (size_t j = 0; j < 4; ++j)
            if (!approx_eq(vals[i][j], o.vals[i][j])) return false;
    return true;
}
// matrix4.rs:155-170
Matrix4 operator*(const Matrix4 &a, const Matrix4 &b)
{
    Matrix4 m;
    for (size_t i = 0; i < 4; ++i)
        for (size_t j = 0; j < 4; ++j) {
            double val = 0.0;
            for (size_t k = 0; k < 4; ++k) val += a.vals[i][k] * b.vals[k][j];
            m.vals[i][j] = val;
        }
    return m;
}
// matrix4.rs:172-193
Tuple operator*(const Matrix4 &a, const Tuple &t)
{
    const double other[4] = { t.x, t.y, t.z, (double)t.v };
    double out[4];
    for (size_t i = 0; i < 4; ++i) {
        double val = 0.0;
        for (size_t k = 0; k < 4; ++k) val += a.vals[i][k] * other[k];
        out[i] = val;
    }
    return { out[0], out[1], out[2], t.v };
}
// transformations.rs:3-61
Matrix4 translation(double x, double y, double z)
{
    Matrix4 m = Matrix4::identity();
    m.vals[0][3] = x; m.vals[1][3] = y; m.vals[2][3] = z;
    return m;
}
Matrix4 scaling(double x, double y, double z)
{
    Matrix4 m;
    m.vals[0][0] = x; m.vals[1][1] = y; m.vals[2][2] = z; m.vals[3][3] = 1.0;
    return m;
}
Matrix4 rotation_x(double rad)
{
    Matrix4 m = Matrix4::identity();
    m.vals[1][1] = std::cos(rad); m.vals[1][2] = -(std::sin(rad));
    m.vals[2][1] = std::sin(rad); m.vals[2][2] = std::cos(rad);
    return m;
}
Matrix4 rotation_y(double rad)
{
    Matrix4 m = Matrix4::identity();
    m.vals[0][0] = std::cos(rad); m.vals[0][2] = std::sin(rad);
    m.vals[2][0] = -(std::sin(rad)); m.vals[2][2] = std::cos(rad);
    return m;
}
Matrix4 rotation_z(double rad)
{
    Matrix4 m = Matrix4::identity();
    m.vals[0][0] = std::cos(rad); m.vals[0][1] = -(std::sin(rad));
    m.vals[1][0] = std::sin(rad); m.vals[1][1] = std::cos(rad);
    return m;
}
Matrix4 shearing(double xy, double xz, double yx, double yz, double zx, double zy)
{
    Matrix4 m = Matrix4::identity();
    m.vals[0][1] = xy; m.vals[0][2] = xz;
    m.vals[1][0] = yx; m.vals[1][2] = yz;
    m.vals[2][0] = zx; m.vals[2][1] = zy;
    return m;
}
// transformations.rs:63-77
Matrix4 view_transform(const Tuple &from, const Tuple &to, const Tuple &up)
{
    const Tuple forward = (to - from).normalize();
    const Tuple left = forward.xprod(up.normalize());
    const Tuple true_up = left.xprod(forward);
    Matrix4 m = Matrix4::identity();
    m.vals[0] = { left.x, left.y, left.z, 0.0 };
    m.vals[1] = { true_up.x, true_up.y, true_up.z, 0.0 };
    m.vals[2] = { -forward.x, -forward.y, -forward.z, 0.0 };
    return m * translation(-from.x, -from.y, -from.z);
}

// ---- rays ---------------------------------------------------------------------------------
Ray::Ray(const Tuple &o, const Tuple &d) : origin(o), direction(d)
{
    if (o.v != 1) throw Panic("Attempted to create ray with vector origin");
    if (d.v != 0) throw Panic("Attempted to create ray with point direction");
}
Tuple Ray::position(double t) const { return origin + (direction * t); }
Ray Ray::transform(const Matrix4 &m) const { return Ray(m * origin, m * direction); }

// ---- patterns, materials, shapes ------------------------------------------------------------
Pattern &Pattern::set_transform(const Matrix4 &m)
{
    transform = m;
    inverse_transform = transform.invert();
    return *this;
}
bool Pattern::operator==(const Pattern &o) const
{
    return pattern_type == o.pattern_type && transform == o.transform && inverse_transform == o.inverse_transform &&
           color1 == o.color1 && color2 == o.color2;
}
Pattern solid() { return Pattern(PatternType::Solid, white(), black()); }
Pattern stripe(const Color &a, const Color &b) { return Pattern(PatternType::Stripe, a, b); }
Pattern gradient(const Color &a, const Color &b) { return Pattern(PatternType::Gradient, a, b); }
Pattern ring(const Color &a, const Color &b) { return Pattern(PatternType::Ring, a, b); }
Pattern checker_3d(const Color &a, const Color &b) { return Pattern(PatternType::Checker3d, a, b); }
Pattern test_pattern() { return Pattern(PatternType::Test, black(), black()); }

bool Material::operator==(const Material &o) const
{
    return color == o.color && ambient == o.ambient && diffuse == o.diffuse && specular == o.specular &&
           shininess == o.shininess && pattern == o.pattern && reflective == o.reflective &&
           transparency == o.transparency && refractive_index == o.refractive_index && casts_shadow == o.casts_shadow;
}
bool Shape::operator==(const Shape &o) const
{
    if (kind != o.kind) return false;
    if (kind == ShapeKind::Cylinder || kind == ShapeKind::Cone) return min == o.min && max == o.max && closed == o.closed;
    return true;
}

Object::Object(const Shape &s) : shape(s) {}
// objects.rs:211-215
void Object::set_transform(const Matrix4 &m)
{
    transform_ = m;
    inverse_ = m.invert();
    inverse_t_ = inverse_.transpose();
}
// objects.rs:217-220
void Object::set_transformation_list(size_t index, double x)
{
    if (index >= 15) throw Panic("index out of bounds: transformations_list");
    list_[index] = x;
    update_transform();
}
// traits.rs:8-33
void Object::update_transform()
{
    const auto &l = list_;
    set_transform(translation(l[0], l[1], l[2]) * scaling(l[3], l[4], l[5]) * rotation_x(l[6]) * rotation_y(l[7]) *
                  rotation_z(l[8]) * shearing(l[9], l[10], l[11], l[12], l[13], l[14]));
}
// traits.rs:35-44
void Object::combine_transforms(const std::array<double, 15> &list)
{
    for (size_t i = 0; i < 15; ++i) {
        if (i <= 2 || i >= 6) list_[i] += list[i];
        else list_[i] *= list[i];
    }
    update_transform();
}
bool Object::operator==(const Object &o) const
{
    return material == o.material && transform_ == o.transform_ && inverse_ == o.inverse_ && inverse_t_ == o.inverse_t_ &&
           shape == o.shape && list_ == o.list_;
}

// ---- groups ----------------------------------------------------------------------------------------
void Group::add_object(const Object &o)
{
    ObjectHolder h;
    h.object = o;
    group_.push_back(h);
}
void Group::add_group(const Group &g)
{
    ObjectHolder h;
    h.is_group = true;
    h.group = std::make_shared<const Group>(g);
    group_.push_back(h);
}
// groups.rs:88-92
void Group::set_transform(const Matrix4 &m)
{
    transform_ = m;
    inverse_ = m.invert();
    inverse_t_ = inverse_.transpose();
}
// groups.rs:94-97
void Group::set_transformation_list(size_t index, double x)
{
    if (index >= 15) throw Panic("index out of bounds: the len is 15 but the index is " + std::to_string(index));
    list_[index] = x;
    update_transform();
}
// traits.rs:8-33
void Group::update_transform()
{
    const auto &l = list_;
    set_transform(translation(l[0], l[1], l[2]) * scaling(l[3], l[4], l[5]) * rotation_x(l[6]) * rotation_y(l[7]) *
                  rotation_z(l[8]) * shearing(l[9], l[10], l[11], l[12], l[13], l[14]));
}

namespace spheres {
Object new_() { return Object(Shape{ ShapeKind::Sphere, 0, 0, false }); }
// spheres.rs:7-19
Object glass_sphere()
{
    Object s = new_();
    s.set_transparency(0.9).set_reflective(0.9).set_refractive_index(1.5).set_color(black()).set_casts_shadow(false);
    s.set_ambient(0.0).set_diffuse(0.0).set_shininess(300.0).set_specular(0.9);
    return s;
}
}  // namespace spheres
namespace planes { Object new_() { return Object(Shape{ ShapeKind::Plane, 0, 0, false }); } }
namespace cubes { Object new_() { return Object(Shape{ ShapeKind::Cube, 0, 0, false }); } }
namespace cylinders {
Object new_(double min, double max, bool closed) { return Object(Shape{ ShapeKind::Cylinder, min, max, closed }); }
Object new_unbounded() { return new_(-1.7976931348623157e308, 1.7976931348623157e308, false); }
}  // namespace cylinders
namespace cones {
Object new_(double min, double max, bool closed) { return Object(Shape{ ShapeKind::Cone, min, max, closed }); }
Object new_unbounded() { return new_(-1.7976931348623157e308, 1.7976931348623157e308, false); }
}  // namespace cones

// ---- device context -------------------------------------------------------------------------
namespace {
[[noreturn]] void raise_rtx(const char *what)
{
    throw Panic(std::string(what) + ": " + rtx_last_error());
}
std::mutex g_ctx_mu;
std::map<int, rtx_context *> g_ctx;

struct SceneHandle {
    rtx_scene *s = nullptr;
    ~SceneHandle() { rtx_scene_destroy(s); }
};
}  // namespace

rtx_context *device_context(int device_id)
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto it = g_ctx.find(device_id);
    if (it != g_ctx.end()) return it->second;
    rtx_context *c = nullptr;
    if (rtx_context_create(device_id, &c) != RTX_OK) raise_rtx("rtx_context_create");
    g_ctx[device_id] = c;
    return c;
}

// ---- world ------------------------------------------------------------------------------------
// world.rs:38-66
World World::new_default()
{
    Object s1 = spheres::new_();
    Material m;
    m.color = color(0.8, 1.0, 0.6);
    m.pattern = solid();
    m.ambient = 0.1;
    m.diffuse = 0.7;
    m.specular = 0.2;
    m.shininess = 200.0;
    s1.set_material(m);
    Object s2 = spheres::new_();
    s2.scale_x(0.5);
    s2.scale_y(0.5);
    s2.scale_z(0.5);
    return World({ s1, s2 }, { Light(point(-10.0, 10.0, -10.0), color(1.0, 1.0, 1.0)) });
}

rtx_scene_desc FlatScene::desc() const
{
    rtx_scene_desc d;
    std::memset(&d, 0, sizeof d);
    d.n_objects = (uint32_t)shape_type.size();
    d.shape_type = shape_type.data();
    d.inverse = inverse.data();
    d.inverse_transposed = inverse_transposed.data();
    d.ymin = ymin.data();
    d.ymax = ymax.data();
    d.closed = closed.data();
    d.color = color.data();
    d.ambient = ambient.data();
    d.diffuse = diffuse.data();
    d.specular = specular.data();
    d.shininess = shininess.data();
    d.reflective = reflective.data();
    d.transparency = transparency.data();
    d.refractive_index = refractive_index.data();
    d.casts_shadow = casts_shadow.data();
    d.pattern_type = pattern_type.data();
    d.pattern_inverse = pattern_inverse.data();
    d.pattern_color1 = pattern_color1.data();
    d.pattern_color2 = pattern_color2.data();
    d.eq_class = eq_class.empty() ? nullptr : eq_class.data();
    d.chain_depth = chain_depth.empty() ? nullptr : chain_depth.data();
    d.chain_first = chain_first.empty() ? nullptr : chain_first.data();
    d.chain_inverse = chain_inverse.empty() ? nullptr : chain_inverse.data();
    d.n_lights = (uint32_t)(light_position.size() / 3);
    d.light_position = light_position.data();
    d.light_intensity = light_intensity.data();
    return d;
}

void World::add_leaves(const Group &g, std::vector<Matrix4> &inverses, std::vector<const std::array<double, 15> *> &lists)
{
    inverses.push_back(g.get_inverse_transform());
    lists.push_back(&g.transformation_list_all());
    for (const ObjectHolder &h : g.elements()) {
        if (h.is_group) {
            add_leaves(*h.group, inverses, lists);
            continue;
        }
        Object merged = h.object;
        for (size_t d = lists.size(); d-- > 0;) merged.combine_transforms(*lists[d]);
        objects_.push_back(merged);
        Chain c;
        c.inverses = inverses;
        c.inverses.push_back(h.object.get_inverse_transform());
        chains_.push_back(c);
    }
    inverses.pop_back();
    lists.pop_back();
}
size_t World::add_group(const Group &g)
{
    const size_t first = objects_.size();
    std::vector<Matrix4> inverses;
    std::vector<const std::array<double, 15> *> lists;
    add_leaves(g, inverses, lists);
    return first;
}

FlatScene World::flatten() const
{
    FlatScene f;
    const size_t n = objects_.size();
    f.inverse.resize(n * 16);
    f.inverse_transposed.resize(n * 16);
    f.pattern_inverse.resize(n * 16);
    bool any_transparent = false;
    for (size_t i = 0; i < n; ++i) {
        const Object &o = objects_[i];
        const Material &m = o.material;
        f.shape_type.push_back((uint8_t)o.shape.kind);
        o.get_inverse_transform().store(&f.inverse[i * 16]);
        o.get_inverse_transform_transposed().store(&f.inverse_transposed[i * 16]);
        f.ymin.push_back(o.shape.min);
        f.ymax.push_back(o.shape.max);
        f.closed.push_back(o.shape.closed ? 1 : 0);
        f.color.insert(f.color.end(), { m.color.red, m.color.green, m.color.blue });
        f.ambient.push_back(m.ambient);
        f.diffuse.push_back(m.diffuse);
        f.specular.push_back(m.specular);
        f.shininess.push_back(m.shininess);
        f.reflective.push_back(m.reflective);
        f.transparency.push_back(m.transparency);
        f.refractive_index.push_back(m.refractive_index);
        f.casts_shadow.push_back(m.casts_shadow ? 1 : 0);
        f.pattern_type.push_back((uint8_t)m.pattern.pattern_type);
        m.pattern.inverse_transform.store(&f.pattern_inverse[i * 16]);
        f.pattern_color1.insert(f.pattern_color1.end(), { m.pattern.color1.red, m.pattern.color1.green, m.pattern.color1.blue });
        f.pattern_color2.insert(f.pattern_color2.end(), { m.pattern.color2.red, m.pattern.color2.green, m.pattern.color2.blue });
        if (m.transparency != 0.0) any_transparent = true;
    }
    // Value-equality classes feed the refraction container list (world.rs:253-285), which
    // only refracted_color and schlick read, and only for a transparent hit object.
    if (any_transparent) {
        f.eq_class.resize(n);
        for (size_t i = 0; i < n; ++i) {
            f.eq_class[i] = (uint32_t)i;
            for (size_t j = 0; j < i; ++j)
                if (objects_[j] == objects_[i]) {
                    f.eq_class[i] = f.eq_class[j];
                    break;
                }
        }
    }
    bool any_chain = false;
    for (const Chain &c : chains_) any_chain = any_chain || !c.inverses.empty();
    if (any_chain) {
        for (size_t i = 0; i < n; ++i) {
            const Chain &c = chains_[i];
            f.chain_depth.push_back(c.inverses.empty() ? 0 : (uint8_t)std::min<size_t>(255, c.inverses.size() - 1));
            f.chain_first.push_back((uint32_t)(f.chain_inverse.size() / 16));
            for (const Matrix4 &m : c.inverses) {
                f.chain_inverse.resize(f.chain_inverse.size() + 16);
                m.store(&f.chain_inverse[f.chain_inverse.size() - 16]);
            }
        }
    }
    for (const Light &l : lights_) {
        f.light_position.insert(f.light_position.end(), { l.position.x, l.position.y, l.position.z });
        f.light_intensity.insert(f.light_intensity.end(), { l.intensity.red, l.intensity.green, l.intensity.blue });
    }
    return f;
}

std::vector<Color> World::color_at(const std::vector<Ray> &rays, uint32_t remaining, int precision, rtx_stats *stats) const
{
    rtx_context *ctx = device_context();
    const FlatScene flat = flatten();
    const rtx_scene_desc d = flat.desc();
    SceneHandle scene;
    if (rtx_scene_upload(ctx, &d, &scene.s) != RTX_OK) raise_rtx("World::color_at");
    std::vector<double> in(rays.size() * 6), out(rays.size() * 3);
    for (size_t i = 0; i < rays.size(); ++i) {
        const Ray &r = rays[i];
        const double v[6] = { r.origin.x, r.origin.y, r.origin.z, r.direction.x, r.direction.y, r.direction.z };
        std::memcpy(&in[i * 6], v, sizeof v);
    }
    if (rtx_trace_rays(ctx, scene.s, in.data(), rays.size(), remaining, precision, out.data(), stats) != RTX_OK)
        raise_rtx("World::color_at");
    std::vector<Color> res(rays.size());
    for (size_t i = 0; i < rays.size(); ++i) res[i] = color(out[i * 3], out[i * 3 + 1], out[i * 3 + 2]);
    return res;
}
Color World::color_at(const Ray &ray, uint32_t remaining) const { return color_at(std::vector<Ray>{ ray }, remaining)[0]; }

// ---- camera -----------------------------------------------------------------------------------
// camera.rs:18-37
Camera::Camera(size_t hsize, size_t vsize, double fov)
    : hsize_(hsize), vsize_(vsize), fov_(fov), components_{ { point(0.0, 0.0, 0.0), point(0.0, 0.0, -1.0), vector(0.0, 1.0, 0.0) } }
{
    initialize();
    update_transformations();
}
// camera.rs:39-55: the aspect ratio and its product with 400 are f32
Camera Camera::new_preview(const Camera &c)
{
    const float ratio = (float)c.hsize_ / (float)c.vsize_;
    const float h = 400.0f * ratio;
    size_t hs = 0;
    if (h == h && h > 0.0f) hs = h >= 18446744073709551615.0f ? SIZE_MAX : (size_t)h;
    Camera n = c;
    n.hsize_ = hs;
    n.vsize_ = 400;
    n.initialize();
    n.update_transformations();
    return n;
}
// camera.rs:87-99
void Camera::initialize()
{
    const double half_view = std::tan(fov_ / 2.0);
    const double aspect = (double)hsize_ / (double)vsize_;
    if (aspect >= 1.0) {
        half_width_ = half_view;
        half_height_ = half_view / aspect;
    } else {
        half_width_ = half_view * aspect;
        half_height_ = half_view;
    }
    pixel_size_ = (half_width_ * 2.0) / (double)hsize_;
}
// camera.rs:101-105
void Camera::set_transform(const Matrix4 &m)
{
    transform_ = m;
    inverse_ = m.invert();
}
// camera.rs:138-145
void Camera::update_transformations() { set_transform(view_transform(components_[0], components_[1], components_[2])); }

rtx_camera_desc Camera::desc() const
{
    rtx_camera_desc d;
    std::memset(&d, 0, sizeof d);
    d.hsize = (uint32_t)hsize_;
    d.vsize = (uint32_t)vsize_;
    inverse_.store(d.inverse_transform);
    d.pixel_size = pixel_size_;
    d.half_width = half_width_;
    d.half_height = half_height_;
    return d;
}

Ray Camera::ray_for_pixel(size_t x, size_t y) const
{
    const rtx_camera_desc d = desc();
    double r[6];
    if (rtx_generate_rays(device_context(), &d, (uint64_t)y * hsize_ + x, 1, r) != RTX_OK) raise_rtx("Camera::ray_for_pixel");
    return Ray(point(r[0], r[1], r[2]), vector(r[3], r[4], r[5]));
}

namespace {
struct Tracker {
    std::atomic<size_t> *counter;
};
void on_progress(uint64_t pixels_done, void *user)
{
    auto *t = static_cast<Tracker *>(user);
    if (t->counter) t->counter->store((size_t)pixels_done, std::memory_order_seq_cst);
}
}  // namespace

void Camera::render_rows(const World &world, uint32_t depth, uint32_t row_begin, uint32_t row_end, double *out,
                         rtx_stats *stats) const
{
    rtx_context *ctx = device_context();
    const FlatScene flat = world.flatten();
    const rtx_scene_desc sd = flat.desc();
    SceneHandle scene;
    if (rtx_scene_upload(ctx, &sd, &scene.s) != RTX_OK) raise_rtx("Camera::render");
    const rtx_camera_desc cd = desc();
    const size_t bytes = (size_t)(row_end - row_begin) * hsize_ * 3 * sizeof(double);
    if (rtx_render(ctx, scene.s, &cd, depth, RTX_F64, row_begin, row_end, out, bytes, nullptr, nullptr, stats) != RTX_OK)
        raise_rtx("Camera::render");
}

Canvas Camera::render(const World &world) const { return parallel_render(world, nullptr, nullptr, RTX_F64); }

Canvas Camera::parallel_render(const World &world, std::atomic<size_t> *tracker, rtx_stats *stats, int precision) const
{
    if (tracker) tracker->store(0, std::memory_order_relaxed);  // camera.rs:173
    rtx_context *ctx = device_context();
    const FlatScene flat = world.flatten();
    const rtx_scene_desc sd = flat.desc();
    SceneHandle scene;
    if (rtx_scene_upload(ctx, &sd, &scene.s) != RTX_OK) raise_rtx("Camera::parallel_render");
    const rtx_camera_desc cd = desc();
    Canvas image(hsize_, vsize_);
    static_assert(sizeof(Color) == 3 * sizeof(double), "Canvas pixels are packed RGB doubles");
    Tracker t{ tracker };
    const size_t n = hsize_ * vsize_ * 3;
    int rc;
    if (precision == RTX_F64) {
        rc = rtx_render(ctx, scene.s, &cd, DEFAULT_RECURSION_DEPTH, RTX_F64, 0, (uint32_t)vsize_, image.pixels.data(),
                        n * sizeof(double), tracker ? on_progress : nullptr, &t, stats);
    } else {
        std::vector<float> tmp(n);
        rc = rtx_render(ctx, scene.s, &cd, DEFAULT_RECURSION_DEPTH, RTX_F32, 0, (uint32_t)vsize_, tmp.data(), n * sizeof(float),
                        tracker ? on_progress : nullptr, &t, stats);
        double *dst = reinterpret_cast<double *>(image.pixels.data());
        for (size_t i = 0; i < n; ++i) dst[i] = (double)tmp[i];
    }
    if (rc != RTX_OK) raise_rtx("Camera::parallel_render");
    return image;
}

Canvas Camera::preview_parallel_render(const World &world) const
{
    Canvas image(hsize_, vsize_);
    render_rows(world, 2, 0, (uint32_t)vsize_, reinterpret_cast<double *>(image.pixels.data()));
    return image;
}

}  // namespace rt
