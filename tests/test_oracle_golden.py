"""Pins the CPU oracle to the reference's own unit tests.

Every case below restates the inputs and expected constants of a `#[test]` in the reference
crate (file:line given per test).  The reference's `assert_eq!` on Tuple / Color / Matrix is
an approximate comparison with tolerance 1e-5 (src/utils.rs:1-11); on bare f64 it is exact.
`APPROX` and `==` below follow that distinction.  The "full precision" values come from an
independent transcription of the same arithmetic (IEEE f64, glibc) and are secondary.
"""
import math

import numpy as np
import pytest

EPS = 1e-5
SQ2 = math.sqrt(2.0)
H = 1.0 / SQ2  # FRAC_1_SQRT_2


def approx(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return a.shape == b.shape and bool(np.all(np.abs(a - b) < EPS))


def ray(o, d):
    return list(o) + list(d)


# ---- src/matrices/tuples.rs:246-303 ------------------------------------------------------------
def test_tuples(orc):
    n = orc.normalize((1.0, 2.0, 3.0))
    assert approx(n, (0.26726, 0.53452, 0.80178))
    assert math.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2] + 0.0) == 1.0  # tuples.rs:268 (exact)
    assert approx(orc.normalize((4.0, 0.0, 0.0)), (1.0, 0.0, 0.0))
    assert approx(orc.reflect_vector((1.0, -1.0, 0.0), (0.0, 1.0, 0.0)), (1.0, 1.0, 0.0))
    assert approx(orc.reflect_vector((0.0, -1.0, 0.0), (H, H, 0.0)), (1.0, 0.0, 0.0))


# ---- src/matrices/matrix4.rs:236-381 -----------------------------------------------------------
def test_matrix_mul_and_tuple(orc):
    api = orc.api
    m = [[1, 2, 3, 4], [5, 6, 7, 8], [9, 8, 7, 6], [5, 4, 3, 2]]
    n = [[-2, 1, 2, 3], [3, 2, 1, -1], [4, 3, 6, 5], [1, 2, 7, 8]]
    q = [[20, 22, 50, 48], [44, 54, 114, 108], [40, 58, 110, 102], [16, 26, 46, 42]]
    assert approx(api.m_mul(m, n), q)
    t = [[1, 2, 3, 4], [2, 4, 4, 2], [8, 6, 4, 1], [0, 0, 0, 1]]
    assert approx(api.m_mul_tuple(t, (1, 2, 3, 1)), (18, 24, 33, 1))
    assert approx(api.m_mul(t, api.m_identity()), t)
    tr = [[0, 9, 3, 0], [9, 8, 0, 8], [1, 8, 5, 3], [0, 0, 5, 8]]
    assert approx(api.m_transpose(tr), np.array(tr, dtype=float).T)


def test_matrix_determinant_and_inverse(orc):
    api = orc.api
    m = [[-2, -8, 3, 5], [-3, 1, 7, 3], [1, 2, -9, 6], [-6, 7, 7, -9]]
    assert api.m_determinant(m) == -4071.0  # matrix4.rs:329 (exact)
    a = [[-5, 2, 6, -8], [1, -5, 1, 8], [7, 7, -6, -7], [1, -3, 7, 4]]
    inv = [[0.21805, 0.45113, 0.24060, -0.04511], [-0.80827, -1.45677, -0.44361, 0.52068],
           [-0.07895, -0.22368, -0.05263, 0.19737], [-0.52256, -0.81391, -0.30075, 0.30639]]
    assert approx(api.m_invert(a), inv)
    a2 = [[3, -9, 7, 3], [3, -8, 2, -9], [-4, 4, 4, 1], [-6, 5, -1, 1]]
    b2 = [[8, 2, 2, 2], [3, -1, 7, 0], [7, 0, 5, 4], [6, -2, 0, 5]]
    c = api.m_mul(a2, b2)
    assert approx(api.m_mul(c, api.m_invert(b2)), a2)
    # non-invertible: |det| < 1e-5 panics (matrix4.rs:100-107)
    with pytest.raises(Exception):
        api.m_invert(np.zeros((4, 4)))
    with pytest.raises(Exception):
        api.m_invert(api.m_scaling(0.01, 0.01, 0.01))  # det = 1e-6


def test_concatenated_transformations(orc):
    api = orc.api
    # Matrix4::identity().rotate_x(pi/2).scale(5,5,5).translate(10,5,7): each helper pre-multiplies
    t = api.m_mul(api.m_rotation_x(math.pi / 2), api.m_identity())
    t = api.m_mul(api.m_scaling(5, 5, 5), t)
    t = api.m_mul(api.m_translation(10, 5, 7), t)
    assert approx(api.m_mul_tuple(t, (1, 0, 1, 1)), (15, 0, 7, 1))


# ---- src/matrices/transformations.rs:83-201 ----------------------------------------------------
def test_transformations(orc):
    api = orc.api
    tr = api.m_translation(5, -3, 2)
    assert approx(api.m_mul_tuple(tr, (-3, 4, 5, 1)), (2, 1, 7, 1))
    assert approx(api.m_mul_tuple(api.m_invert(tr), (-3, 4, 5, 1)), (-8, 7, 3, 1))
    assert approx(api.m_mul_tuple(tr, (-3, 4, 5, 0)), (-3, 4, 5, 0))
    sc = api.m_scaling(2, 3, 4)
    assert approx(api.m_mul_tuple(sc, (-4, 6, 8, 1)), (-8, 18, 32, 1))
    assert approx(api.m_mul_tuple(api.m_invert(sc), (-4, 6, 8, 0)), (-2, 2, 2, 0))
    assert approx(api.m_mul_tuple(api.m_rotation_x(math.pi / 4), (0, 1, 0, 1)), (0, H, H, 1))
    assert approx(api.m_mul_tuple(api.m_rotation_x(math.pi / 2), (0, 1, 0, 1)), (0, 0, 1, 1))
    assert approx(api.m_mul_tuple(api.m_invert(api.m_rotation_x(math.pi / 4)), (0, 1, 0, 1)), (0, H, -H, 1))
    assert approx(api.m_mul_tuple(api.m_rotation_y(math.pi / 4), (0, 0, 1, 1)), (H, 0, H, 1))
    assert approx(api.m_mul_tuple(api.m_rotation_y(math.pi / 2), (0, 0, 1, 1)), (1, 0, 0, 1))
    assert approx(api.m_mul_tuple(api.m_rotation_z(math.pi / 4), (0, 1, 0, 1)), (-H, H, 0, 1))
    assert approx(api.m_mul_tuple(api.m_rotation_z(math.pi / 2), (0, 1, 0, 1)), (-1, 0, 0, 1))
    p = (2, 3, 4, 1)
    expect = [(5, 3, 4), (6, 3, 4), (2, 5, 4), (2, 7, 4), (2, 3, 6), (2, 3, 7)]
    for k in range(6):
        s = [0.0] * 6
        s[k] = 1.0
        assert approx(api.m_mul_tuple(api.m_shearing(*s), p)[:3], expect[k])


def test_view_transform(orc):
    api = orc.api
    assert approx(api.m_view_transform((0, 0, 0), (0, 0, -1), (0, 1, 0)), api.m_identity())
    assert approx(api.m_view_transform((0, 0, 0), (0, 0, 1), (0, 1, 0)), api.m_scaling(-1, 1, -1))
    assert approx(api.m_view_transform((0, 0, 8), (0, 0, 0), (0, 1, 0)), api.m_translation(0, 0, -8))
    m = [[-0.50709, 0.50709, 0.67612, -2.36643], [0.76772, 0.60609, 0.12122, -2.82843],
         [-0.35857, 0.59761, -0.71714, 0.0], [0.0, 0.0, 0.0, 1.0]]
    assert approx(api.m_view_transform((1, 3, 2), (4, -2, 8), (1, 1, 0)), m)


# ---- src/rays.rs:78-106 ------------------------------------------------------------------------
def test_ray_transform_via_intersections(orc):
    # scaled sphere (spheres.rs:108-123): t = 3, 7 (exact); translated sphere misses (:125-135)
    s = orc.api.sphere().set_transform(orc.api.m_scaling(2, 2, 2))
    assert orc.intersect(s, ray((0, 0, -5), (0, 0, 1))) == [3.0, 7.0]
    s = orc.api.sphere().set_transform(orc.api.m_translation(5, 0, 0))
    assert orc.intersect(s, ray((0, 0, -5), (0, 0, 1))) == []


# ---- src/shapes/spheres.rs:60-187 --------------------------------------------------------------
def test_sphere(orc):
    s = orc.api.sphere()
    assert orc.intersect(s, ray((0, 0, -5), (0, 0, 1))) == [4.0, 6.0]
    assert orc.intersect(s, ray((0, 1, -5), (0, 0, 1))) == [5.0, 5.0]
    assert orc.intersect(s, ray((0, 2, -5), (0, 0, 1))) == []
    assert orc.intersect(s, ray((0, 0, 0), (0, 0, 1))) == [-1.0, 1.0]
    assert approx(orc.normal_at(s, (1, 0, 0)), (1, 0, 0))
    assert approx(orc.normal_at(s, (0, 1, 0)), (0, 1, 0))
    assert approx(orc.normal_at(s, (0, 0, 1)), (0, 0, 1))
    f = math.sqrt(3.0) / 3.0
    n = orc.normal_at(s, (f, f, f))
    assert approx(n, (f, f, f)) and approx(n, orc.normalize(n))
    s.set_transform(orc.api.m_translation(0, 1, 0))
    assert approx(orc.normal_at(s, (0, 1.70711, -H)), (0, H, -H))
    s.set_transform(orc.api.m_mul(orc.api.m_scaling(1, 0.5, 1), orc.api.m_rotation_z(math.pi / 5)))
    assert approx(orc.normal_at(s, (0, H, -H)), (0, 0.97014, -0.24254))


def test_glass_sphere_preset(orc):
    # spheres.rs:7-19 observed through refraction bookkeeping: ri 1.5, transparency 0.9
    g = orc.api.glass()
    c = orc.prepare_computations([-1.0, 1.0], [g, g], 1, ray((0, 0, 0), (0, 1, 0)))
    assert (c.n1, c.n2) == (1.5, 1.0)


# ---- src/shapes/planes.rs:23-57 ----------------------------------------------------------------
def test_plane(orc):
    p = orc.api.plane()
    assert orc.local_intersect(p, ray((0, 10, 0), (0, 0, 1))) == []
    assert orc.local_intersect(p, ray((0, 0, 0), (0, 0, 1))) == []
    assert orc.local_intersect(p, ray((0, 1, 0), (0, -1, 0))) == [1.0]
    assert orc.local_intersect(p, ray((0, -1, 0), (0, 1, 0))) == [1.0]
    assert approx(orc.normal_at(p, (3, 0, -7)), (0, 1, 0))


# ---- src/shapes/cubes.rs:69-178 ----------------------------------------------------------------
def test_cube(orc):
    c = orc.api.cube()
    hits = [((5, 0.5, 0), (-1, 0, 0)), ((-5, 0.5, 0), (1, 0, 0)), ((0.5, 5, 0), (0, -1, 0)), ((0.5, -5, 0), (0, 1, 0)),
            ((0.5, 0, 5), (0, 0, -1)), ((0.5, 0, -5), (0, 0, 1))]
    for o, d in hits:
        assert orc.local_intersect(c, ray(o, d)) == [4.0, 6.0]
    assert orc.local_intersect(c, ray((0, 0.5, 0), (0, 0, 1))) == [-1.0, 1.0]
    misses = [((-2, 0, 0), (0.2673, 0.5345, 0.8018)), ((0, -2, 0), (0.8018, 0.2673, 0.5345)),
              ((0, 0, -2), (0.5345, 0.8018, 0.2673)), ((2, 0, 2), (0, 0, -1)), ((0, 2, 2), (0, -1, 0)),
              ((2, 2, 0), (-1, 0, 0))]
    for o, d in misses:
        assert orc.local_intersect(c, ray(o, d)) == []
    normals = [((1, 0.5, -0.8), (1, 0, 0)), ((-1, -0.2, 0.9), (-1, 0, 0)), ((-0.4, 1, -0.1), (0, 1, 0)),
               ((0.3, -1, -0.7), (0, -1, 0)), ((-0.6, 0.3, 1), (0, 0, 1)), ((0.4, 0.4, -1), (0, 0, -1)),
               ((1, 1, 1), (1, 0, 0)), ((-1, -1, -1), (-1, 0, 0))]
    for p, n in normals:
        assert approx(orc.normal_at(c, p), n)


# ---- src/shapes/cylinders.rs:111-240 -----------------------------------------------------------
def test_cylinder(orc):
    api = orc.api
    cyl = api.cylinder()
    assert orc.intersect(cyl, ray((1, 0, 0), (0, 1, 0))) == []
    assert orc.intersect(cyl, ray((0, 0, 0), (0, 1, 0))) == []
    assert orc.intersect(cyl, ray((0, 0, -5), orc.normalize((1, 1, 1)))) == []
    assert orc.intersect(cyl, ray((1, 0, -5), (0, 0, 1))) == [5.0, 5.0]
    assert orc.intersect(cyl, ray((0, 0, -5), (0, 0, 1))) == [4.0, 6.0]
    xs = orc.intersect(cyl, ray((0.5, 0, -5), orc.normalize((0.1, 1, 1))))
    assert approx(xs, (6.80798, 7.08872))
    for p, n in [((1, 0, 0), (1, 0, 0)), ((0, 5, -1), (0, 0, -1)), ((0, -2, 1), (0, 0, 1)), ((-1, 1, 0), (-1, 0, 0))]:
        assert approx(orc.normal_at(cyl, p), n)
    tr = api.cylinder(1.0, 2.0, False)
    counts = [((0, 1.5, 0), orc.normalize((0.1, 1, 0)), 0), ((0, 3, -5), (0, 0, 1), 0), ((0, 0, -5), (0, 0, 1), 0),
              ((0, 2, -5), (0, 0, 1), 0), ((0, 1, -5), (0, 0, 1), 0), ((0, 1.5, -2), (0, 0, 1), 2)]
    for o, d, k in counts:
        assert len(orc.intersect(tr, ray(o, d))) == k
    cl = api.cylinder(1.0, 2.0, True)
    caps = [((0, 3, 0), (0, -1, 0)), ((0, 3, -2), orc.normalize((0, -1, 2))), ((0, 4, -2), orc.normalize((0, -1, 1))),
            ((0, 0, -2), orc.normalize((0, 1, 2))), ((0, -1, -2), orc.normalize((0, 1, 1)))]
    for o, d in caps:
        assert len(orc.intersect(cl, ray(o, d))) == 2
    for p, n in [((0, 1, 0), (0, -1, 0)), ((0.5, 1, 0), (0, -1, 0)), ((0, 1, 0.5), (0, -1, 0)), ((0, 2, 0), (0, 1, 0)),
                 ((0.5, 2, 0), (0, 1, 0)), ((0, 2, 0.5), (0, 1, 0))]:
        assert approx(orc.normal_at(cl, p), n)


# ---- src/shapes/cones.rs:133-192 ---------------------------------------------------------------
def test_cone(orc):
    api = orc.api
    cone = api.cone()
    assert orc.intersect(cone, ray((0, 0, -5), (0, 0, 1))) == [5.0, 5.0]
    assert approx(orc.intersect(cone, ray((0, 0, -5), orc.normalize((1, 1, 1)))), (8.66025, 8.66025))
    assert approx(orc.intersect(cone, ray((1, 1, -5), orc.normalize((-0.5, -1, 1)))), (4.55006, 49.44994))
    xs = orc.intersect(cone, ray((0, 0, -1), orc.normalize((0, 1, 1))))
    assert len(xs) == 1 and approx(xs, (0.35355,))
    capped = api.cone(-0.5, 0.5, True)
    assert len(orc.intersect(capped, ray((0, 0, -5), (0, 1, 0)))) == 0
    assert len(orc.intersect(capped, ray((0, 0, -0.25), orc.normalize((0, 1, 1))))) == 2
    assert len(orc.intersect(capped, ray((0, 0, -0.25), (0, 1, 0)))) == 4
    # cones.rs:181-192 calls the local normal directly: (0,0,0), (1,-sqrt2,1), (-1,1,0); through
    # Object::normal_at the same vectors come back normalised (the first is 0/0 = NaN)
    assert approx(orc.normal_at(cone, (1, 1, 1)), orc.normalize((1, -SQ2, 1)))
    assert approx(orc.normal_at(cone, (-1, -1, 0)), orc.normalize((-1, 1, 0)))


# ---- src/surfaces/patterns.rs:178-270, src/scenes/lights.rs:131-148 ----------------------------
def test_patterns(orc):
    api = orc.api
    W, B = (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)
    st = api.sphere().set_pattern(orc.scene_api.STRIPE, W, B)
    for p, c in [((0, 0, 0), W), ((0, 1, 0), W), ((0, 2, 0), W), ((0, 0, 1), W), ((0, 0, 2), W), ((1, 0, 0), B),
                 ((-1, 0, 0), B), ((-1.1, 0, 0), W)]:
        assert approx(orc.pattern_at(st, p), c)
    gr = api.sphere().set_pattern(orc.scene_api.GRADIENT, W, B)
    for x, v in [(0, 1.0), (0.25, 0.75), (0.5, 0.5), (0.75, 0.25)]:
        assert approx(orc.pattern_at(gr, (x, 0, 0)), (v, v, v))
    rg = api.sphere().set_pattern(orc.scene_api.RING, W, B)
    for p, c in [((0, 0, 0), W), ((1, 0, 0), B), ((0, 0, 1), B), ((0.708, 0, 0.708), B)]:
        assert approx(orc.pattern_at(rg, p), c)
    ck = api.sphere().set_pattern(orc.scene_api.CHECKER3D, W, B)
    for p, c in [((0, 0, 0), W), ((0.99, 0, 0), W), ((1.01, 0, 0), B), ((0, 0.99, 0), W), ((0, 1.01, 0), B),
                 ((0, 0, 0.99), W), ((0, 0, 1.01), B)]:
        assert approx(orc.pattern_at(ck, p), c)
    o = api.sphere().set_pattern(orc.scene_api.STRIPE, W, B).set_pattern_transform(api.m_scaling(2, 2, 2))
    assert approx(orc.pattern_at_object(o, (1.5, 0, 0)), W)
    o = api.sphere().set_transform(api.m_scaling(2, 2, 2)).set_pattern(orc.scene_api.STRIPE, W, B)
    o.set_pattern_transform(api.m_translation(0.5, 0, 0))
    assert approx(orc.pattern_at_object(o, (2.5, 0, 0)), W)
    o = api.sphere().set_transform(api.m_scaling(2, 2, 2)).set_pattern(orc.scene_api.STRIPE, W, B)
    assert approx(orc.pattern_at_object(o, (1.5, 0, 0)), W)
    # negative odd cells: -1 % 2 == -1 in Rust, so floor(x) = -1 selects the second colour
    assert approx(orc.pattern_at(st, (-0.5, 0, 0)), B)
    assert approx(orc.pattern_at(ck, (-0.5, 0.5, 0.5)), B)


# ---- src/scenes/lights.rs:84-148 ---------------------------------------------------------------
def test_lighting(orc):
    s = orc.api.sphere()
    W = (1.0, 1.0, 1.0)
    pos, n = (0, 0, 0), (0, 0, -1)
    assert approx(orc.lighting(s, (0, 0, -10), W, pos, (0, 0, -1), n, False), (1.9, 1.9, 1.9))
    assert approx(orc.lighting(s, (0, 0, -10), W, pos, (0, H, -H), n, False), (1.0, 1.0, 1.0))
    assert approx(orc.lighting(s, (0, 10, -10), W, pos, (0, 0, -1), n, False), (0.7364, 0.7364, 0.7364))
    assert approx(orc.lighting(s, (0, 10, -10), W, pos, (0, -H, -H), n, False), (1.6364, 1.6364, 1.6364))
    assert approx(orc.lighting(s, (0, 0, 10), W, pos, (0, 0, -1), n, False), (0.1, 0.1, 0.1))
    assert approx(orc.lighting(s, (0, 0, -10), W, pos, (0, 0, -1), n, True), (0.1, 0.1, 0.1))
    p = orc.api.sphere().set_specular(0).set_diffuse(0).set_ambient(1).set_pattern(orc.scene_api.STRIPE, W, (0, 0, 0))
    assert approx(orc.lighting(p, (0, 0, -10), W, (0.9, 0, 0), (0, 0, -1), n, False), W)
    assert approx(orc.lighting(p, (0, 0, -10), W, (1.1, 0, 0), (0, 0, -1), n, False), (0, 0, 0))


# ---- src/scenes/world.rs:295-732 ---------------------------------------------------------------
def test_default_world_intersect(orc):
    w = orc.api.default_world()
    ts, objs = orc.world_intersect(w, ray((0, 0, -5), (0, 0, 1)))
    assert ts == [4.0, 4.5, 5.5, 6.0]  # world.rs:301-305 (exact)
    assert objs == [0, 1, 1, 0]


def test_prepare_computations(orc):
    api = orc.api
    s = api.sphere()
    c = orc.prepare_computations([1.0], [s], 0, ray((0, 0, 0), (0, 0, 1)))  # world.rs:308-320
    assert approx(c.point[:], (0, 0, 1)) and approx(c.eyev[:], (0, 0, -1)) and c.inside == 1
    assert approx(c.normalv[:], (0, 0, -1))
    sh = api.sphere().set_transform(api.m_translation(0, 0, 1))  # world.rs:322-334
    c = orc.prepare_computations([5.0], [sh], 0, ray((0, 0, -5), (0, 0, 1)))
    assert c.over_point[2] < -EPS / 2 and c.point[2] > c.over_point[2]
    pl = api.plane()  # world.rs:445-453
    c = orc.prepare_computations([SQ2], [pl], 0, ray((0, 1, -1), (0, -H, H)))
    assert approx(c.reflectv[:], (0, H, H))


def test_shade_hit_and_color_at(orc):
    api = orc.api
    w = api.default_world()
    r = ray((0, 0, -5), (0, 0, 1))
    c = orc.prepare_computations([4.0], [w.object(0)], 0, r)
    assert approx(orc.shade_hit(w, w.object(0), c), (0.38066, 0.47583, 0.2855))  # world.rs:336-347
    col = orc.color_at(w, [r])[0]
    assert approx(col, (0.38066, 0.47583, 0.2855))  # world.rs:395-401
    assert np.allclose(col, (0.3806609553101071, 0.47582619413763383, 0.2854957164825803), rtol=0, atol=1e-15)
    assert approx(orc.color_at(w, [ray((0, 0, -5), (0, 1, 0))])[0], (0, 0, 0))  # world.rs:387-393
    # ray inside, light at (0, 0.25, 0) (world.rs:349-363)
    w2 = api.default_world().clear_lights().add_light((0, 0.25, 0))
    r2 = ray((0, 0, 0), (0, 0, 1))
    c = orc.prepare_computations([0.5], [w2.object(1)], 0, r2)
    got = orc.shade_hit(w2, w2.object(1), c)
    assert approx(got, (0.90498, 0.90498, 0.90498))
    assert abs(got[0] - 0.9049812520679432) < 1e-15
    # in shadow (world.rs:365-385)
    w3 = api.world().add_light((0, 0, -10))
    w3.add_object(api.sphere())
    w3.add_object(api.sphere())
    w3.object(1).set_transform(api.m_mul(api.m_translation(0, 0, 10), api.m_identity()))
    c = orc.prepare_computations([4.0], [w3.object(1)], 0, ray((0, 0, 5), (0, 0, 1)))
    assert approx(orc.shade_hit(w3, w3.object(1), c), (0.1, 0.1, 0.1))
    # colour of the inner sphere seen from between the spheres (world.rs:403-429)
    w4 = api.default_world()
    w4.object(0).set_ambient(1.0)
    w4.object(1).set_ambient(1.0)
    assert approx(orc.color_at(w4, [ray((0, 0, 0.75), (0, 0, -1))])[0], (1.0, 1.0, 1.0))


def test_is_shadowed(orc):
    w = orc.api.default_world()  # world.rs:431-443
    assert not orc.is_shadowed(w, (0, 10, 0))
    assert not orc.is_shadowed(w, (-20, 20, -20))
    assert not orc.is_shadowed(w, (-2, 2, -2))
    assert orc.is_shadowed(w, (10, -10, 10))


def _world_with_floor(orc, **floor):
    api = orc.api
    w = api.default_world()
    shape = api.plane().set_transform(api.m_translation(0, -1, 0))
    for k, v in floor.items():
        getattr(shape, "set_" + k)(v)
    w.add_object(shape)
    return w, shape


def test_reflection(orc):
    api = orc.api
    w = api.default_world()  # world.rs:455-470
    w.object(1).set_ambient(1.0)
    c = orc.prepare_computations([1.0], [w.object(1)], 0, ray((0, 0, 0), (0, 0, 1)))
    assert approx(orc.reflected_color(w, w.object(1), c), (0, 0, 0))
    w, shape = _world_with_floor(orc, reflective=0.5)  # world.rs:472-524
    r = ray((0, 0, -3), (0, -H, H))
    c = orc.prepare_computations([SQ2], [shape], 0, r)
    got = orc.reflected_color(w, shape, c)
    assert approx(got, (0.19033, 0.23791, 0.14275))
    assert np.allclose(got, (0.19033220149513294, 0.23791525186891616, 0.1427491511213497), rtol=0, atol=1e-15)
    got = orc.shade_hit(w, shape, c)
    assert approx(got, (0.87676, 0.92434, 0.82918))
    assert np.allclose(got, (0.8767572837020907, 0.9243403340758739, 0.8291742333283074), rtol=0, atol=1e-15)
    assert approx(orc.reflected_color(w, shape, c, remaining=0), (0, 0, 0))


def test_refractive_indices(orc):
    api = orc.api  # world.rs:526-567
    a = api.glass().set_transform(api.m_scaling(2, 2, 2)).set_refractive_index(1.5)
    b = api.glass().set_transform(api.m_translation(0, 0, -0.25)).set_refractive_index(2.0)
    c = api.glass().set_transform(api.m_translation(0, 0, 0.25)).set_refractive_index(2.5)
    ts = [2.0, 2.75, 3.25, 4.75, 5.25, 6.0]
    objs = [a, b, c, b, c, a]
    expect = [(1.0, 1.5), (1.5, 2.0), (2.0, 2.5), (2.5, 2.5), (2.5, 1.5), (1.5, 1.0)]
    for i, (n1, n2) in enumerate(expect):
        comps = orc.prepare_computations(ts, objs, i, ray((0, 0, -4), (0, 0, 1)))
        assert (comps.n1, comps.n2) == (n1, n2)


def test_refraction(orc):
    api = orc.api
    w = api.default_world()
    s0 = w.object(0)
    r = ray((0, 0, -5), (0, 0, 1))
    c = orc.prepare_computations([4.0, 6.0], [s0, s0], 0, r)
    assert approx(orc.refracted_color(w, s0, c, 5), (0, 0, 0))  # opaque, world.rs:569-583
    assert approx(orc.refracted_color(w, s0, c, 0), (0, 0, 0))  # depth 0, world.rs:585-599
    c = orc.prepare_computations([-H, H], [s0, s0], 1, ray((0, 0, H), (0, 1, 0)))
    assert approx(orc.refracted_color(w, s0, c, 5), (0, 0, 0))  # TIR, world.rs:601-615
    # world.rs:617-645
    w = api.default_world()
    w.object(0).set_ambient(1.0).set_pattern(orc.scene_api.TEST, (0, 0, 0), (0, 0, 0))
    w.object(1).set_transparency(1.0).set_refractive_index(1.5)
    s1, s2 = w.object(0), w.object(1)
    c = orc.prepare_computations([-0.9899, -0.4899, 0.4899, 0.9899], [s1, s2, s2, s1], 2, ray((0, 0, 0.1), (0, 1, 0)))
    got = orc.refracted_color(w, s2, c, 5)
    assert approx(got, (0.0, 0.99887, 0.04722))
    assert np.allclose(got, (0.0, 0.9988745506795582, 0.04721898034382347), rtol=0, atol=1e-15)


def _floor_and_ball(orc, **floor):
    api = orc.api
    w = api.default_world()
    fl = api.plane().set_transform(api.m_translation(0, -1, 0))
    for k, v in floor.items():
        getattr(fl, "set_" + k)(v)
    w.add_object(fl)
    ball = api.sphere().set_color(1, 0, 0).set_ambient(0.5).set_transform(api.m_translation(0, -3.5, -0.5))
    w.add_object(ball)
    return w, fl


def test_shade_hit_refraction_and_schlick(orc):
    r = ray((0, 0, -3), (0, -H, H))
    w, fl = _floor_and_ball(orc, transparency=0.5, refractive_index=1.5)  # world.rs:647-671
    c = orc.prepare_computations([SQ2], [fl], 0, r)
    got = orc.shade_hit(w, fl, c, 5)
    assert approx(got, (0.93642, 0.68642, 0.68642))
    assert np.allclose(got, (0.9364250822069577, 0.6864250822069577, 0.6864250822069577), rtol=0, atol=1e-15)
    w, fl = _floor_and_ball(orc, reflective=0.5, transparency=0.5, refractive_index=1.5)  # world.rs:698-722
    c = orc.prepare_computations([SQ2], [fl], 0, r)
    got = orc.shade_hit(w, fl, c, 5)
    assert approx(got, (0.93391, 0.69643, 0.69243))
    assert np.allclose(got, (0.9339149012949264, 0.696434003918285, 0.6924304352337541), rtol=0, atol=1e-15)


def test_schlick(orc):
    g = orc.api.glass()
    c = orc.prepare_computations([-H, H], [g, g], 1, ray((0, 0, H), (0, 1, 0)))
    assert orc.schlick(c) == 1.0  # world.rs:673-684 (exact)
    c = orc.prepare_computations([-1.0, 1.0], [g, g], 1, ray((0, 0, 0), (0, 1, 0)))
    assert abs(orc.schlick(c) - 0.04) < EPS  # world.rs:686-696
    assert abs(orc.schlick(c) - 0.04000000000000001) < 1e-17
    c = orc.prepare_computations([1.8589], [g], 0, ray((0, 0.99, -2), (0, 0, 1)))
    assert abs(orc.schlick(c) - 0.48873) < EPS  # world.rs:724-732
    assert abs(orc.schlick(c) - 0.4887308101221217) < 1e-15


# ---- src/scenes/camera.rs:229-302 --------------------------------------------------------------
def test_camera(orc):
    api = orc.api
    assert abs(api.camera(200, 125, math.pi / 2).get()["pixel_size"] - 0.01) < EPS
    assert abs(api.camera(125, 200, math.pi / 2).get()["pixel_size"] - 0.01) < EPS
    c = api.camera(201, 101, math.pi / 2)
    assert approx(orc.ray_for_pixel(c, 100, 50), (0, 0, 0, 0, 0, -1))
    r = orc.ray_for_pixel(c, 0, 0)
    assert approx(r, (0, 0, 0, 0.66519, 0.33259, -0.66851))
    assert np.allclose(r[3:], (0.6651864261194508, 0.3325932130597254, -0.6685123582500481), rtol=0, atol=1e-15)
    c.set_transform(api.m_mul(api.m_rotation_y(math.pi / 4), api.m_mul(api.m_translation(0, -2, 5), api.m_identity())))
    assert approx(orc.ray_for_pixel(c, 100, 50), (0, 2, -5, H, 0, -H))


def test_render_world(orc):
    api = orc.api
    w = api.default_world()
    c = api.camera(11, 11, math.pi / 2)
    c.set_transform(api.m_view_transform((0, 0, -5), (0, 0, 0), (0, 1, 0)))
    img = orc.render_rows(c, w)  # camera.rs:275-287
    assert approx(img[5, 5], (0.38066, 0.47583, 0.2855))
    cnt = [0, 0]
    par = orc.parallel_render(c, w, n_threads=3, counters=cnt)  # camera.rs:289-302
    assert np.array_equal(par, img)
    assert cnt[0] == 121 and cnt[1] == int(np.count_nonzero(img.sum(axis=2)))


def test_preview_camera(orc):
    c = orc.api.camera(1000, 800, math.pi / 2)
    p = c.preview().get()  # camera.rs:39-55
    assert (p["hsize"], p["vsize"]) == (500, 400)
    p = orc.api.camera(1920, 1080, 1.0).preview().get()
    assert (p["hsize"], p["vsize"]) == (711, 400)


# ---- src/scenes/canvas.rs:52-71, 84-86, 126-146 ------------------------------------------------
def test_canvas_quantisation(orc):
    v = np.array([1.5, 0.0, 0.0, 0.0, 0.5, 0.0, -0.5, 0.0, 1.0])
    assert list(orc.quantize_ppm(v)) == [255, 0, 0, 0, 127, 0, 0, 0, 255]
    assert list(orc.quantize_u8(v)) == [255, 0, 0, 0, 127, 0, 0, 0, 255]
    assert list(orc.quantize_u8(np.array([0.999, 1.0, float("nan"), 0.0039063, 0.0039062]))) == [255, 255, 0, 1, 0]


# ---- transformation list (src/shapes/traits.rs:8-98) -------------------------------------------
def test_transformation_list(orc):
    api = orc.api
    o = api.sphere().translate(1, 2, 3).scale(2, 2, 2).rotate_y(0.3).shear_xy(0.1)
    t, inv, inv_t = o.matrices()
    expect = api.m_chain(api.m_translation(1, 2, 3), api.m_scaling(2, 2, 2), api.m_rotation_x(0.0), api.m_rotation_y(0.3),
                         api.m_rotation_z(0.0), api.m_shearing(0.1, 0, 0, 0, 0, 0))
    assert np.array_equal(t, expect)
    assert np.array_equal(inv, api.m_invert(expect)) and np.array_equal(inv_t, inv.T)
    z = api.sphere().scale_x(0.0)  # a zero scale is replaced by EPSILON (traits.rs:58-64) -> det 1e-5 is invertible
    assert z.matrices()[0][0, 0] == 1e-5
    with pytest.raises(Exception):
        api.sphere().scale(0.0, 0.0, 1.0)  # det 1e-10: the reference panics in invert()
