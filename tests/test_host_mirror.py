"""CPU-side checks of the product library: it loads, exports every symbol of the C ABI, refuses
to render without a GPU, and its host-side scene construction (matrix composition, cofactor
inverse, camera set-up, flattening) is bit-identical to the oracle's."""
import ctypes
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:rtx|rt)_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rta):
    for header, prefix in [("rtx.h", "rtx_"), ("rt_scene.h", "rt_")]:
        names = [n for n in _declared(header) if n.startswith(prefix) and not n.endswith("_fn")]
        assert len(names) >= 13
        for n in names:
            assert hasattr(rta.lib, n), "%s is declared in include/%s but not exported" % (n, header)
    assert sorted(rta.RTX_SYMBOLS) == [n for n in _declared("rtx.h") if n != "rtx_progress_fn"]


def test_struct_layouts_match_header(rta):
    assert ctypes.sizeof(rta.Stats) == 8 + 8 + 64 + 24 + 8 + 8
    assert ctypes.sizeof(rta.Probe) == 8 * 21 + 16
    assert ctypes.sizeof(rta.CameraDesc) == 8 + 128 + 24


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_no_cpu_fallback(rta):
    """Without a device every render entry point fails loudly; nothing routes to the CPU."""
    if _has_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(rta.RtxError, match="no HIP device"):
        rta.Context()
    w = rta.api.default_world()
    cam = rta.api.camera(8, 8, 1.0)
    with pytest.raises(rta.Panic, match="no HIP device"):
        rta.render_rows(cam, w)
    with pytest.raises(rta.Panic, match="no HIP device"):
        rta.color_at(w, [[0, 0, -5, 0, 0, 1]])


def test_matrix_helpers_bit_identical(orc, rta):
    rng = np.random.default_rng(3)
    a, b = rta.api, orc.api
    for _ in range(200):
        v = rng.uniform(-3, 3, size=6)
        for name, args in [("m_translation", v[:3]), ("m_scaling", v[:3]), ("m_rotation_x", v[:1]), ("m_rotation_y", v[1:2]),
                           ("m_rotation_z", v[2:3]), ("m_shearing", v)]:
            assert np.array_equal(getattr(a, name)(*args), getattr(b, name)(*args))
        m = rng.uniform(-2, 2, size=(4, 4))
        n = rng.uniform(-2, 2, size=(4, 4))
        assert np.array_equal(a.m_mul(m, n), b.m_mul(m, n))
        assert a.m_determinant(m) == b.m_determinant(m)
        if abs(b.m_determinant(m)) >= 1e-5:
            assert np.array_equal(a.m_invert(m), b.m_invert(m))
        t = np.append(rng.uniform(-5, 5, size=3), rng.integers(0, 2))
        assert np.array_equal(a.m_mul_tuple(m, t), b.m_mul_tuple(m, t))
        f, to, up = rng.uniform(-5, 5, size=(3, 3))
        assert np.array_equal(a.m_view_transform(f, to, up), b.m_view_transform(f, to, up))
    with pytest.raises(rta.Panic):
        a.m_invert(np.zeros((4, 4)))


def _same_object(o1, o2):
    for m1, m2 in zip(o1.matrices(), o2.matrices()):
        if not np.array_equal(m1, m2):
            return False
    return np.array_equal(o1.pattern_inverse(), o2.pattern_inverse())


def test_object_transforms_bit_identical(orc, rta):
    rng = np.random.default_rng(5)
    for _ in range(100):
        vals = rng.uniform(-2, 2, size=15)
        vals[3:6] = rng.uniform(0.3, 2.0, size=3)
        g = rta.api.cube()
        for slot in rng.permutation(15):
            g._op(int(slot), vals[slot])
        # the list is recomposed as T*S*Rx*Ry*Rz*Shear on every call, whatever the call order
        o = orc.api.cube()
        for slot in range(15):
            o._op(slot, vals[slot])
        assert _same_object(g, o)
    # explicit matrix + pattern transform
    m = rta.api.m_chain(rta.api.m_translation(1, 2, 3), rta.api.m_rotation_y(0.3), rta.api.m_scaling(2, 0.5, 1))
    g = rta.api.sphere().set_transform(m).set_pattern(rta.STRIPE).set_pattern_transform(rta.api.m_scaling(0.1, 2, 3))
    o = orc.api.sphere().set_transform(m).set_pattern(rta.STRIPE).set_pattern_transform(orc.api.m_scaling(0.1, 2, 3))
    assert _same_object(g, o)
    with pytest.raises(rta.Panic):
        rta.api.sphere().set_transform(np.zeros((4, 4)))
    assert rta.api.sphere().scale_x(0.0).matrices()[0][0, 0] == 1e-5


def test_object_equality_mirror(orc, rta):
    for api in (rta.api, orc.api):
        a, b = api.sphere(), api.sphere()
        assert a.equals(b)
        b.set_color(1.0, 1.0, 1.0 - 5e-6)  # colours compare with tolerance 1e-5 (colors.rs:83-92)
        assert a.equals(b)
        b.set_ambient(0.1 + 1e-12)  # scalars compare exactly (derived PartialEq on f64)
        assert not a.equals(b)
        assert not api.cylinder(0, 1, True).equals(api.cylinder(0, 1, False))
        assert not api.sphere().equals(api.cube())


def test_camera_mirror_bit_identical(orc, rta):
    rng = np.random.default_rng(9)
    for h, v, fov in [(1000, 800, math.pi / 2), (125, 200, 1.0), (3840, 2160, math.pi / 3), (7680, 4320, 0.7)]:
        f, to = rng.uniform(-5, 5, size=(2, 3))
        g = rta.api.camera(h, v, fov).set_view(f, to, (0, 1, 0)).get()
        o = orc.api.camera(h, v, fov).set_view(f, to, (0, 1, 0)).get()
        for k in g:
            assert np.array_equal(g[k], o[k]), k
        gp = rta.api.camera(h, v, fov).preview().get()
        op = orc.api.camera(h, v, fov).preview().get()
        assert (gp["hsize"], gp["vsize"], gp["pixel_size"]) == (op["hsize"], op["vsize"], op["pixel_size"])


def test_flatten_matches_oracle_objects(orc, rta):
    from ray_tracer_amd import scenes
    gw, _ = scenes.glass_scene(rta.api, 64, 48, n_lights=4)
    ow, _ = scenes.glass_scene(orc.api, 64, 48, n_lights=4)
    flat = rta.flatten(gw)
    n = len(gw)
    assert n == len(ow) == 8
    inv = flat["inverse"].reshape(n, 4, 4)
    inv_t = flat["inverse_transposed"].reshape(n, 4, 4)
    pinv = flat["pattern_inverse"].reshape(n, 4, 4)
    for i in range(n):
        _, oi, oit = ow.object(i).matrices()
        assert np.array_equal(inv[i], oi) and np.array_equal(inv_t[i], oit)
        assert np.array_equal(pinv[i], ow.object(i).pattern_inverse())
    assert list(flat["shape_type"]) == [1, 1, 0, 0, 0, 2, 3, 4]
    assert list(flat["casts_shadow"]) == [1, 1, 0, 0, 1, 0, 1, 1]
    assert list(flat["closed"]) == [0, 0, 0, 0, 0, 0, 1, 1]
    assert flat["light_position"].size == 12  # all four lights travel; shading reads lights[0] only
    assert list(flat["eq_class"]) == list(range(n))
    # value-identical objects share a class (refraction containers alias them, world.rs:267-276)
    w = rta.api.world()
    w.add_object(rta.api.glass())
    w.add_object(rta.api.sphere())
    w.add_object(rta.api.glass())
    w.add_light((0, 0, 0))
    assert list(rta.flatten(w)["eq_class"]) == [0, 1, 0]


def test_default_world_mirror(orc, rta):
    g, o = rta.api.default_world(), orc.api.default_world()
    assert len(g) == len(o) == 2
    for i in range(2):
        assert _same_object(g.object(i), o.object(i))
    flat = rta.flatten(g)
    assert np.array_equal(flat["color"], [0.8, 1.0, 0.6, 1.0, 1.0, 1.0])
    assert np.array_equal(flat["light_position"], [-10.0, 10.0, -10.0])


def test_scene_generators_are_deterministic(orc):
    from ray_tracer_amd import scenes
    a, _ = scenes.random_scene(orc.api, n_objects=50)
    b, _ = scenes.random_scene(orc.api, n_objects=50)
    for i in range(52):
        assert _same_object(a.object(i), b.object(i))
    # SplitMix64 reference vector (seed 1234567)
    rng = scenes.SplitMix64(1234567)
    assert [rng.next_u64() for _ in range(3)] == [6457827717110365317, 3203168211198807973, 9817491932198370423]


def test_unknown_shape_is_refused(rta):
    """rt_object_new hands out no object for a shape tag outside sphere .. cone."""
    api = rta.api
    assert not api.object_new(5, 0.0, 0.0, 0)
    assert not api.object_new(-1, 0.0, 0.0, 0)
    assert b"unknown shape" in api.last_panic()
