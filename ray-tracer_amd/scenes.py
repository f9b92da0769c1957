"""The benchmark and parity scenes, built through the reference's scene API.

Every builder takes a `SceneAPI` (product library or, in the test-suite, the oracle) and
returns `(world, camera)`; the same description therefore feeds both sides of a parity test.
Random scenes draw from a self-implemented SplitMix64 so that every harness gets identical
objects.
"""
import math

from .scene_api import CHECKER3D, GRADIENT, RING, STRIPE

MASK64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.state = seed & MASK64

    def next_u64(self):
        self.state = (self.state + 0x9E3779B97F4A7C15) & MASK64
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * ((self.next_u64() >> 11) * (1.0 / (1 << 53)))


def _camera(api, hsize, vsize, fov, frm, to, up=(0.0, 1.0, 0.0)):
    cam = api.camera(hsize, vsize, fov)
    cam.set_view(frm, to, up)
    return cam


def default_scene(api, hsize=1000, vsize=800):
    """S1: World::new_default (src/scenes/world.rs:38-66) seen by the GUI's default camera
    (src/app.rs:24-25, 62-64): FOV pi/2 from (0,1.5,-5) to (0,1.5,0)."""
    world = api.default_world()
    cam = _camera(api, hsize, vsize, math.pi / 2.0, (0.0, 1.5, -5.0), (0.0, 1.5, 0.0))
    return world, cam


def spheres_on_plane(api, hsize=3840, vsize=2160):
    """S1b: a floor plane and three spheres, one light; every pixel hits, shadows matter."""
    w = api.world()
    w.add_object(api.plane())
    w.add_object(api.sphere().translate(-0.5, 1.0, 0.5).set_color(0.1, 1.0, 0.5).set_diffuse(0.7).set_specular(0.3))
    w.add_object(api.sphere().translate(1.5, 0.5, -0.5).scale(0.5, 0.5, 0.5).set_color(0.5, 1.0, 0.1)
                 .set_diffuse(0.7).set_specular(0.3))
    w.add_object(api.sphere().translate(-1.5, 0.33, -0.75).scale(0.33, 0.33, 0.33).set_color(1.0, 0.8, 0.1)
                 .set_diffuse(0.7).set_specular(0.3))
    w.add_light((-10.0, 10.0, -10.0))
    cam = _camera(api, hsize, vsize, math.pi / 3.0, (0.0, 1.5, -5.0), (0.0, 1.0, 0.0))
    return w, cam


def glass_scene(api, hsize=3840, vsize=2160, n_lights=1):
    """S3 (and S5 with n_lights=4): checker floor, glass sphere with an air bubble, mirror
    sphere, glass cube, closed cylinder and cone with patterns, striped back wall."""
    w = api.world()
    floor = api.plane().set_pattern(CHECKER3D, (0.9, 0.9, 0.9), (0.15, 0.15, 0.2)).set_reflective(0.3).set_specular(0.1)
    w.add_object(floor)
    wall = api.plane().rotate_x(math.pi / 2.0).translate_z(6.0)
    wall.set_pattern(STRIPE, (0.8, 0.3, 0.3), (0.9, 0.9, 0.8))
    wall.set_pattern_transform(api.m_chain(api.m_rotation_y(math.pi / 5.0), api.m_scaling(0.5, 0.5, 0.5)))
    wall.set_specular(0.0)
    w.add_object(wall)
    w.add_object(api.glass().translate(0.0, 1.0, 0.0))
    bubble = api.sphere().translate(0.0, 1.0, 0.0).scale(0.5, 0.5, 0.5)
    bubble.set_color(0.0, 0.0, 0.0).set_ambient(0.0).set_diffuse(0.0).set_transparency(1.0).set_refractive_index(1.0)
    bubble.set_casts_shadow(False)
    w.add_object(bubble)
    mirror = api.sphere().translate(-2.4, 0.8, 1.2).scale(0.8, 0.8, 0.8)
    mirror.set_color(0.05, 0.05, 0.1).set_diffuse(0.2).set_reflective(0.9).set_shininess(300.0)
    w.add_object(mirror)
    # the cube floats 0.05 above the floor: a bottom face coplanar with the plane would make the
    # winner of every ray through it a matter of rounding
    cube = api.cube().translate(2.3, 0.65, 0.8).scale(0.6, 0.6, 0.6).rotate_y(0.6)
    cube.set_color(0.05, 0.1, 0.05).set_diffuse(0.1).set_ambient(0.05).set_transparency(0.8).set_reflective(0.5)
    cube.set_refractive_index(1.5).set_casts_shadow(False)
    w.add_object(cube)
    cyl = api.cylinder(0.0, 1.5, True).translate(-1.2, 0.0, -1.6).scale(0.4, 1.0, 0.4)
    cyl.set_pattern(RING, (0.2, 0.4, 0.9), (0.9, 0.9, 0.2)).set_pattern_transform(api.m_scaling(0.3, 0.3, 0.3))
    w.add_object(cyl)
    cone = api.cone(-1.0, 0.0, True).translate(1.3, 1.0, -1.8).scale(0.5, 1.0, 0.5)
    cone.set_pattern(GRADIENT, (0.9, 0.2, 0.6), (0.2, 0.9, 0.9))
    cone.set_pattern_transform(api.m_chain(api.m_translation(0.5, 0.0, 0.0), api.m_scaling(2.0, 1.0, 1.0)))
    cone.set_reflective(0.2)
    w.add_object(cone)
    w.add_light((-10.0, 10.0, -10.0))
    extra = [((10.0, 8.0, -6.0), (0.6, 0.6, 0.6)), ((0.0, 12.0, 4.0), (0.4, 0.4, 0.5)), ((-4.0, 3.0, -9.0), (0.3, 0.2, 0.2))]
    for pos, inten in extra[: max(0, n_lights - 1)]:
        w.add_light(pos, inten)  # ignored by shading: the reference reads lights[0] only (world.rs:123,147)
    cam = _camera(api, hsize, vsize, math.pi / 3.0, (0.0, 2.2, -6.5), (0.0, 1.0, 0.0))
    return w, cam


def random_scene(api, n_objects=10000, hsize=3840, vsize=2160, seed=0xC0FFEE, light=(-30.0, 60.0, -40.0), no_shadow_every=0,
                 glass_every=0):
    """S4: n random spheres / cubes / closed cylinders plus a checker floor and a striped back
    wall.  Scales stay >= 0.3 so every transform has |det| >= 1e-5 (the reference panics
    below that, src/matrices/matrix4.rs:100-107).  Variants for tests: another light position;
    every no_shadow_every-th object casting no shadow; every glass_every-th object made of glass
    (transparency 0.9, reflective 0.9, refractive index 1.5, as spheres::glass_sphere has it)."""
    rng = SplitMix64(seed)
    w = api.world()
    for index in range(n_objects):
        kind = rng.uniform()
        if kind < 0.5:
            o = api.sphere()
        elif kind < 0.75:
            o = api.cube()
        else:
            o = api.cylinder(0.0, 1.0, True)
        cx, cy, cz = rng.uniform(-40.0, 40.0), rng.uniform(0.5, 25.0), rng.uniform(-5.0, 120.0)
        s = rng.uniform(0.3, 1.2)
        rx, ry, rz = (rng.uniform(0.0, 2.0 * math.pi) for _ in range(3))
        col = (rng.uniform(), rng.uniform(), rng.uniform())
        refl = rng.uniform() < 0.1
        o.set_transform(api.m_chain(api.m_translation(cx, cy, cz), api.m_scaling(s, s, s), api.m_rotation_x(rx),
                                    api.m_rotation_y(ry), api.m_rotation_z(rz)))
        o.set_color(*col)
        if refl:
            o.set_reflective(0.3)
        if no_shadow_every and index % no_shadow_every == 0:
            o.set_casts_shadow(False)
        if glass_every and index % glass_every == 1:
            o.set_transparency(0.9).set_reflective(0.9).set_refractive_index(1.5)
        w.add_object(o)
    w.add_object(api.plane().set_pattern(CHECKER3D, (0.85, 0.85, 0.85), (0.2, 0.2, 0.25))
                 .set_pattern_transform(api.m_scaling(2.0, 2.0, 2.0)).set_specular(0.1))
    wall = api.plane().rotate_x(math.pi / 2.0).translate_z(130.0)
    wall.set_pattern(STRIPE, (0.3, 0.4, 0.8), (0.8, 0.8, 0.9)).set_pattern_transform(api.m_scaling(3.0, 3.0, 3.0))
    wall.set_specular(0.0)
    w.add_object(wall)
    w.add_light(light)
    cam = _camera(api, hsize, vsize, math.pi / 3.0, (0.0, 12.0, -30.0), (0.0, 8.0, 40.0))
    return w, cam


def grouped_scene(api, hsize=160, vsize=90, nested=True):
    """Objects inside Groups (src/shapes/groups.rs): a transformed group holding one of each shape,
    a nested group two levels down, a child positioned with set_transform (which the merged copy
    used for shading discards, src/shapes/traits.rs:35-44), next to plain objects."""
    w = api.world()
    w.add_object(api.plane().set_pattern(CHECKER3D, (0.9, 0.9, 0.9), (0.2, 0.2, 0.25)).set_reflective(0.2))
    g = api.group().translate(0.5, 0.0, 1.0).rotate_y(0.4).scale(1.2, 1.2, 1.2)
    g.add_object(api.sphere().translate(-1.5, 1.0, 0.0).set_color(0.9, 0.3, 0.2).set_reflective(0.3))
    g.add_object(api.cube().translate(1.2, 0.5, 0.3).scale(0.5, 0.5, 0.5).rotate_y(0.3)
                 .set_pattern(STRIPE, (0.2, 0.6, 0.9), (0.9, 0.9, 0.3)).set_pattern_transform(api.m_scaling(0.3, 0.3, 0.3)))
    g.add_object(api.cylinder(0.0, 1.2, True).translate(0.0, 0.0, -1.5).scale(0.4, 1.0, 0.4).set_color(0.3, 0.8, 0.4))
    g.add_object(api.cone(-1.0, 0.0, True).translate(2.4, 1.0, -0.8).scale(0.5, 1.0, 0.5)
                 .set_pattern(GRADIENT, (0.9, 0.2, 0.6), (0.2, 0.9, 0.9)))
    if nested:
        inner = api.group().translate(0.0, 1.6, 0.0).rotate_z(0.3).shear_xy(0.15)
        inner.add_object(api.glass().scale(0.6, 0.6, 0.6))
        inner.add_object(api.sphere().scale(0.25, 0.25, 0.25).set_color(0.1, 0.1, 0.9))
        deepest = api.group().translate_x(1.1).scale(0.5, 0.5, 0.5)
        deepest.add_object(api.cube().rotate_x(0.5).set_color(0.9, 0.8, 0.1).set_transparency(0.4).set_refractive_index(1.3))
        inner.add_group(deepest)
        g.add_group(inner)
    moved = api.sphere().set_color(0.8, 0.8, 0.8)
    moved.set_transform(api.m_chain(api.m_translation(-2.8, 0.5, 1.5), api.m_scaling(0.5, 0.5, 0.5)))
    g.add_object(moved)
    w.add_group(g)
    w.add_object(api.sphere().translate(-3.0, 0.7, -1.0).scale(0.7, 0.7, 0.7).set_color(0.6, 0.2, 0.7))
    solo = api.group()  # an untransformed group: merged lists equal the child's own
    solo.add_object(api.cube().translate(3.2, 0.4, 2.0).scale(0.4, 0.4, 0.4).set_reflective(0.5))
    w.add_group(solo)
    w.add_light((-6.0, 8.0, -7.0))
    cam = _camera(api, hsize, vsize, math.pi / 3.0, (0.0, 3.0, -8.0), (0.0, 1.0, 0.0))
    return w, cam


def random_scene_with_glass(api, n_objects=10000, hsize=3840, vsize=2160, **kw):
    """The random scene with every 400th object made of glass (25 of 10 000)."""
    return random_scene(api, n_objects=n_objects, hsize=hsize, vsize=vsize, glass_every=400, **kw)


SCENES = {
    "default": default_scene,
    "spheres_on_plane": spheres_on_plane,
    "glass": glass_scene,
    "random": random_scene,
    "random_glass": random_scene_with_glass,
    "grouped": grouped_scene,
}
