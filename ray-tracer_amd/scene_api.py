"""Thin ctypes wrappers over a C library that exports the reference's scene-building API.

Two libraries export that API with identical signatures and a different prefix: the product
library (``rt_``, include/rt_scene.h) and the test oracle (``orc_``, oracle/rt_oracle.h).
`SceneAPI` binds the shared subset (objects, materials, patterns, worlds, cameras, matrix
helpers) so that one scene description can be built in either; the package binds it to the
product library only, the test-suite also binds it to the oracle.

Names follow the reference crate (src/shapes/objects.rs, src/shapes/traits.rs,
src/scenes/world.rs, src/scenes/camera.rs).
"""
import ctypes as C

import numpy as np

SPHERE, PLANE, CUBE, CYLINDER, CONE = range(5)
SOLID, STRIPE, GRADIENT, RING, CHECKER3D, TEST = range(6)
AMBIENT, DIFFUSE, SPECULAR, SHININESS, REFLECTIVE, TRANSPARENCY, REFRACTIVE_INDEX = range(7)
F64, F32 = 0, 1

F64_MAX = 1.7976931348623157e308

_dp = C.POINTER(C.c_double)


def _vec(v, n):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
    if a.size != n:
        raise ValueError("expected %d values, got %d" % (n, a.size))
    return a


def _ptr(a):
    return a.ctypes.data_as(_dp)


class Panic(RuntimeError):
    """The reference would have panicked (or the device failed)."""


class SceneAPI:
    def __init__(self, lib, prefix):
        self.lib = lib
        self.prefix = prefix
        f = self._fn
        vp = C.c_void_p
        d = C.c_double
        f("last_panic", C.c_char_p, [])
        for name in ("identity",):
            f(name, None, [_dp])
        f("translation", None, [d, d, d, _dp])
        f("scaling", None, [d, d, d, _dp])
        for name in ("rotation_x", "rotation_y", "rotation_z"):
            f(name, None, [d, _dp])
        f("shearing", None, [d] * 6 + [_dp])
        f("mat4_mul", None, [_dp, _dp, _dp])
        f("mat4_transpose", None, [_dp, _dp])
        f("mat4_determinant", d, [_dp])
        f("mat4_invert", C.c_int, [_dp, _dp])
        f("mat4_mul_tuple", None, [_dp, _dp, _dp])
        f("view_transform", C.c_int, [_dp, _dp, _dp, _dp])
        f("object_new", vp, [C.c_int, d, d, C.c_int])
        f("glass_sphere", vp, [])
        f("object_clone", vp, [vp])
        f("object_free", None, [vp])
        f("object_set_transform", C.c_int, [vp, _dp])
        f("object_set_transformation_list", C.c_int, [vp, C.c_int, d])
        f("object_transform_op", C.c_int, [vp, C.c_int, d])
        f("object_set_color", None, [vp, d, d, d])
        f("object_set_scalar", None, [vp, C.c_int, d])
        f("object_set_casts_shadow", None, [vp, C.c_int])
        f("object_set_pattern", None, [vp, C.c_int, _dp, _dp])
        f("object_set_pattern_transform", C.c_int, [vp, _dp])
        f("object_get_matrices", None, [vp, _dp, _dp, _dp])
        f("object_get_pattern_inverse", None, [vp, _dp])
        f("object_equal", C.c_int, [vp, vp])
        f("group_new", vp, [])
        f("group_free", None, [vp])
        f("group_add_object", C.c_int, [vp, vp])
        f("group_add_group", C.c_int, [vp, vp])
        f("group_n_elements", C.c_int, [vp])
        f("group_set_transform", C.c_int, [vp, _dp])
        f("group_set_transformation_list", C.c_int, [vp, C.c_int, d])
        f("group_transform_op", C.c_int, [vp, C.c_int, d])
        f("group_get_matrices", None, [vp, _dp, _dp, _dp])
        f("world_add_group", C.c_int, [vp, vp])
        f("world_object_chain", C.c_int, [vp, C.c_int, _dp, C.c_int])
        f("world_new", vp, [])
        f("world_default", vp, [])
        f("world_free", None, [vp])
        f("world_add_object", C.c_int, [vp, vp])
        f("world_add_light", None, [vp, _dp, _dp])
        f("world_clear_lights", None, [vp])
        f("world_n_objects", C.c_int, [vp])
        f("world_object", vp, [vp, C.c_int])
        f("camera_new", vp, [C.c_uint64, C.c_uint64, d])
        f("camera_new_preview", vp, [vp])
        f("camera_free", None, [vp])
        f("camera_set_transform", C.c_int, [vp, _dp])
        f("camera_set_view", C.c_int, [vp, _dp, _dp, _dp])
        f("camera_get", None, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _dp, _dp, _dp, _dp])

    def _fn(self, name, restype, argtypes):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = restype
        fn.argtypes = argtypes
        setattr(self, name, fn)
        return fn

    def check(self, status):
        if status != 0:
            raise Panic(self.last_panic().decode("utf-8", "replace"))

    # ---- matrices ------------------------------------------------------------------------
    def _m(self, fn, *args):
        out = np.empty(16)
        fn(*args, _ptr(out))
        return out.reshape(4, 4)

    def m_identity(self):
        return self._m(self.identity)

    def m_translation(self, x, y, z):
        return self._m(self.translation, x, y, z)

    def m_scaling(self, x, y, z):
        return self._m(self.scaling, x, y, z)

    def m_rotation_x(self, r):
        return self._m(self.rotation_x, r)

    def m_rotation_y(self, r):
        return self._m(self.rotation_y, r)

    def m_rotation_z(self, r):
        return self._m(self.rotation_z, r)

    def m_shearing(self, *s):
        return self._m(self.shearing, *s)

    def m_mul(self, a, b):
        a, b = _vec(a, 16), _vec(b, 16)
        return self._m(self.mat4_mul, _ptr(a), _ptr(b))

    def m_chain(self, *ms):
        """Left-to-right product a * b * c ... as the Rust `*` chain evaluates it."""
        r = np.asarray(ms[0], dtype=np.float64)
        for m in ms[1:]:
            r = self.m_mul(r, m)
        return r

    def m_transpose(self, a):
        a = _vec(a, 16)
        return self._m(self.mat4_transpose, _ptr(a))

    def m_determinant(self, a):
        a = _vec(a, 16)
        return self.mat4_determinant(_ptr(a))

    def m_invert(self, a):
        a = _vec(a, 16)
        out = np.empty(16)
        self.check(self.mat4_invert(_ptr(a), _ptr(out)))
        return out.reshape(4, 4)

    def m_mul_tuple(self, m, t):
        m, t = _vec(m, 16), _vec(t, 4)
        out = np.empty(4)
        self.mat4_mul_tuple(_ptr(m), _ptr(t), _ptr(out))
        return out

    def m_view_transform(self, frm, to, up):
        a, b, c = _vec(frm, 3), _vec(to, 3), _vec(up, 3)
        out = np.empty(16)
        self.check(self.view_transform(_ptr(a), _ptr(b), _ptr(c), _ptr(out)))
        return out.reshape(4, 4)

    # ---- constructors (src/shapes/*.rs) ------------------------------------------------------
    def sphere(self):
        return Object(self, self.object_new(SPHERE, 0.0, 0.0, 0))

    def glass(self):
        return Object(self, self.glass_sphere())

    def plane(self):
        return Object(self, self.object_new(PLANE, 0.0, 0.0, 0))

    def cube(self):
        return Object(self, self.object_new(CUBE, 0.0, 0.0, 0))

    def cylinder(self, ymin=-F64_MAX, ymax=F64_MAX, closed=False):
        return Object(self, self.object_new(CYLINDER, ymin, ymax, int(closed)))

    def cone(self, ymin=-F64_MAX, ymax=F64_MAX, closed=False):
        return Object(self, self.object_new(CONE, ymin, ymax, int(closed)))

    def group(self):
        return Group(self, self.group_new())

    def world(self):
        return World(self, self.world_new())

    def default_world(self):
        return World(self, self.world_default())

    def camera(self, hsize, vsize, fov):
        h = self.camera_new(hsize, vsize, fov)
        if not h:
            raise Panic(self.last_panic().decode("utf-8", "replace"))
        return Camera(self, h)


class Object:
    """src/shapes/objects.rs `Object` plus the `Transformable` helpers of src/shapes/traits.rs."""

    def __init__(self, api, handle, owned=True):
        self.api = api
        self.h = handle
        self.owned = owned

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            self.api.object_free(self.h)
            self.h = None

    def clone(self):
        return Object(self.api, self.api.object_clone(self.h))

    def set_transform(self, m):
        m = _vec(m, 16)
        self.api.check(self.api.object_set_transform(self.h, _ptr(m)))
        return self

    def set_transformation_list(self, index, x):
        self.api.check(self.api.object_set_transformation_list(self.h, index, x))
        return self

    def _op(self, slot, x):
        self.api.check(self.api.object_transform_op(self.h, slot, float(x)))
        return self

    def translate_x(self, v): return self._op(0, v)
    def translate_y(self, v): return self._op(1, v)
    def translate_z(self, v): return self._op(2, v)
    def scale_x(self, v): return self._op(3, v)
    def scale_y(self, v): return self._op(4, v)
    def scale_z(self, v): return self._op(5, v)
    def rotate_x(self, v): return self._op(6, v)
    def rotate_y(self, v): return self._op(7, v)
    def rotate_z(self, v): return self._op(8, v)
    def shear_xy(self, v): return self._op(9, v)
    def shear_xz(self, v): return self._op(10, v)
    def shear_yx(self, v): return self._op(11, v)
    def shear_yz(self, v): return self._op(12, v)
    def shear_zx(self, v): return self._op(13, v)
    def shear_zy(self, v): return self._op(14, v)

    def translate(self, x, y, z):
        return self.translate_x(x).translate_y(y).translate_z(z)

    def scale(self, x, y, z):
        return self.scale_x(x).scale_y(y).scale_z(z)

    def set_color(self, r, g, b):
        self.api.object_set_color(self.h, r, g, b)
        return self

    def _scalar(self, field, v):
        self.api.object_set_scalar(self.h, field, float(v))
        return self

    def set_ambient(self, v): return self._scalar(AMBIENT, v)
    def set_diffuse(self, v): return self._scalar(DIFFUSE, v)
    def set_specular(self, v): return self._scalar(SPECULAR, v)
    def set_shininess(self, v): return self._scalar(SHININESS, v)
    def set_reflective(self, v): return self._scalar(REFLECTIVE, v)
    def set_transparency(self, v): return self._scalar(TRANSPARENCY, v)
    def set_refractive_index(self, v): return self._scalar(REFRACTIVE_INDEX, v)

    def set_casts_shadow(self, v):
        self.api.object_set_casts_shadow(self.h, int(bool(v)))
        return self

    def set_pattern(self, ptype, c1=(1.0, 1.0, 1.0), c2=(0.0, 0.0, 0.0)):
        a, b = _vec(c1, 3), _vec(c2, 3)
        self.api.object_set_pattern(self.h, ptype, _ptr(a), _ptr(b))
        return self

    def set_pattern_transform(self, m):
        m = _vec(m, 16)
        self.api.check(self.api.object_set_pattern_transform(self.h, _ptr(m)))
        return self

    def matrices(self):
        t, i, it = np.empty(16), np.empty(16), np.empty(16)
        self.api.object_get_matrices(self.h, _ptr(t), _ptr(i), _ptr(it))
        return t.reshape(4, 4), i.reshape(4, 4), it.reshape(4, 4)

    def pattern_inverse(self):
        m = np.empty(16)
        self.api.object_get_pattern_inverse(self.h, _ptr(m))
        return m.reshape(4, 4)

    def equals(self, other):
        return bool(self.api.object_equal(self.h, other.h))


class Group:
    """src/shapes/groups.rs `Group` with the `Transformable` helpers of src/shapes/traits.rs.
    Adding an object or a group stores a copy, as moving the value does in Rust."""

    def __init__(self, api, handle):
        self.api = api
        self.h = handle

    def __del__(self):
        if getattr(self, "h", None):
            self.api.group_free(self.h)
            self.h = None

    def add_object(self, obj):
        self.api.group_add_object(self.h, obj.h)
        return self

    def add_group(self, group):
        self.api.group_add_group(self.h, group.h)
        return self

    def __len__(self):
        return self.api.group_n_elements(self.h)

    def set_transform(self, m):
        m = _vec(m, 16)
        self.api.check(self.api.group_set_transform(self.h, _ptr(m)))
        return self

    def set_transformation_list(self, index, x):
        self.api.check(self.api.group_set_transformation_list(self.h, index, x))
        return self

    def _op(self, slot, x):
        self.api.check(self.api.group_transform_op(self.h, slot, float(x)))
        return self

    def translate_x(self, v): return self._op(0, v)
    def translate_y(self, v): return self._op(1, v)
    def translate_z(self, v): return self._op(2, v)
    def scale_x(self, v): return self._op(3, v)
    def scale_y(self, v): return self._op(4, v)
    def scale_z(self, v): return self._op(5, v)
    def rotate_x(self, v): return self._op(6, v)
    def rotate_y(self, v): return self._op(7, v)
    def rotate_z(self, v): return self._op(8, v)
    def shear_xy(self, v): return self._op(9, v)
    def shear_xz(self, v): return self._op(10, v)
    def shear_yx(self, v): return self._op(11, v)
    def shear_yz(self, v): return self._op(12, v)
    def shear_zx(self, v): return self._op(13, v)
    def shear_zy(self, v): return self._op(14, v)

    def translate(self, x, y, z):
        return self.translate_x(x).translate_y(y).translate_z(z)

    def scale(self, x, y, z):
        return self.scale_x(x).scale_y(y).scale_z(z)

    def matrices(self):
        t, i, it = np.empty(16), np.empty(16), np.empty(16)
        self.api.group_get_matrices(self.h, _ptr(t), _ptr(i), _ptr(it))
        return t.reshape(4, 4), i.reshape(4, 4), it.reshape(4, 4)


class World:
    """src/scenes/world.rs `World`."""

    def __init__(self, api, handle):
        self.api = api
        self.h = handle

    def __del__(self):
        if getattr(self, "h", None):
            self.api.world_free(self.h)
            self.h = None

    def add_object(self, obj):
        return self.api.world_add_object(self.h, obj.h)

    def add_group(self, group):
        """World::add_group: returns the index of the group's first object in the flattened list."""
        first = self.api.world_add_group(self.h, group.h)
        if first < 0:
            raise Panic(self.api.last_panic().decode("utf-8", "replace"))
        return first

    def chain(self, index):
        """Inverse transforms of the groups around object `index`, outermost first: (depth, 4, 4)."""
        buf = np.zeros(16 * 16)
        depth = self.api.world_object_chain(self.h, index, _ptr(buf), 16)
        return buf[: 16 * max(depth, 0)].reshape(-1, 4, 4)

    def add_light(self, position, intensity=(1.0, 1.0, 1.0)):
        p, i = _vec(position, 3), _vec(intensity, 3)
        self.api.world_add_light(self.h, _ptr(p), _ptr(i))
        return self

    def clear_lights(self):
        self.api.world_clear_lights(self.h)
        return self

    def __len__(self):
        return self.api.world_n_objects(self.h)

    def object(self, index):
        """Mutable view of a stored object; valid until the next add_object."""
        h = self.api.world_object(self.h, index)
        if not h:
            raise IndexError(index)
        return Object(self.api, h, owned=False)


class Camera:
    """src/scenes/camera.rs `Camera`."""

    def __init__(self, api, handle):
        self.api = api
        self.h = handle

    def __del__(self):
        if getattr(self, "h", None):
            self.api.camera_free(self.h)
            self.h = None

    def preview(self):
        h = self.api.camera_new_preview(self.h)
        if not h:
            raise Panic(self.api.last_panic().decode("utf-8", "replace"))
        return type(self)(self.api, h)

    def set_transform(self, m):
        m = _vec(m, 16)
        self.api.check(self.api.camera_set_transform(self.h, _ptr(m)))
        return self

    def set_view(self, frm, to, up):
        a, b, c = _vec(frm, 3), _vec(to, 3), _vec(up, 3)
        self.api.check(self.api.camera_set_view(self.h, _ptr(a), _ptr(b), _ptr(c)))
        return self

    def get(self):
        hs, vs = C.c_uint64(), C.c_uint64()
        ps, hw, hh = C.c_double(), C.c_double(), C.c_double()
        inv = np.empty(16)
        self.api.camera_get(self.h, C.byref(hs), C.byref(vs), C.byref(ps), C.byref(hw), C.byref(hh), _ptr(inv))
        return dict(hsize=hs.value, vsize=vs.value, pixel_size=ps.value, half_width=hw.value,
                    half_height=hh.value, inverse=inv.reshape(4, 4))

    @property
    def hsize(self):
        return self.get()["hsize"]

    @property
    def vsize(self):
        return self.get()["vsize"]
