"""Camera rays and their shadow rays in scenes with the bounding-sphere hierarchy are walked a wave
at a time: 64 lanes test 64 spheres against a cone around the tile's rays (from the camera), or
around the hit points as the light sees them.  What the cone rules out no ray could have hit, so
the frames are those of the same scene rendered without any bounding-sphere test, and those of the
oracle's World::color_at (src/scenes/world.rs:98-119: every object, sorted, first t >= 0).  The
views here are the ones that stress the cone: the camera inside the cloud of objects (half of them
behind it), beyond it looking back, a field of view so wide that a tile's rays span a right angle,
frames smaller than a tile and with ragged edges, a light in the middle of the objects."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VIEWS = {
    "front": dict(fov=math.pi / 3.0, frm=(0.0, 12.0, -30.0), to=(0.0, 8.0, 40.0)),
    "inside": dict(fov=math.pi / 2.0, frm=(3.0, 9.0, 55.0), to=(-10.0, 8.0, 80.0)),
    "far_side": dict(fov=math.pi / 3.0, frm=(5.0, 30.0, 128.0), to=(0.0, 5.0, 40.0)),
    "from_above": dict(fov=math.pi / 4.0, frm=(0.0, 90.0, 50.0), to=(0.0, 0.0, 50.5), up=(0.0, 0.0, 1.0)),
}


def _world(api, n_objects, **variant):
    from ray_tracer_amd import scenes
    return scenes.random_scene(api, n_objects=n_objects, hsize=8, vsize=8, **variant)[0]


def _camera(api, W, H, view):
    cam = api.camera(W, H, view["fov"])
    cam.set_view(view["frm"], view["to"], view.get("up", (0.0, 1.0, 0.0)))
    return cam


def _frame(rta, ctx, scene, cd, precision):
    out = np.empty((cd.vsize, cd.hsize, 3), dtype=np.float64 if precision == rta.F64 else np.float32)
    st = rta.Stats()
    rta._check(rta.rtx_render(ctx.h, scene.h, cd, 5, precision, 0, cd.vsize, out.ctypes.data, out.nbytes, None, None, st))
    return out, st.as_dict()


def _both(rta, world, cams, precision=None):
    """every camera's frame with the hierarchy (threshold 0) and without (never)"""
    precision = rta.F64 if precision is None else precision
    flat = rta.flatten(world)
    got = {}
    for threshold in (0, 0xffffffff):
        ctx = rta.Context()
        ctx.set_cull_threshold(threshold)
        scene = rta.Scene(ctx, flat)
        assert scene.features()["bounding_sphere_tests"] == (threshold == 0)
        for name, cam in cams.items():
            got[(threshold, name)] = _frame(rta, ctx, scene, rta.camera_desc(cam), precision)
        scene.close()
        ctx.close()
    return got


@pytest.mark.parametrize("n_objects", [700, 2500])
def test_views_of_a_cloud_of_objects(rta, n_objects):
    """2500 objects: several groups per shape table, so the walk pops groups and clusters in fours."""
    world = _world(rta.api, n_objects)
    cams = {name: _camera(rta.api, 160, 90, v) for name, v in VIEWS.items()}
    got = _both(rta, world, cams)
    for name in cams:
        (a, sa), (b, sb) = got[(0, name)], got[(0xffffffff, name)]
        assert sa["rays_per_level"] == sb["rays_per_level"], name
        assert sa["shadow_rays"] == sb["shadow_rays"], name
        assert np.abs(a - b).max() <= 1e-12, name
        assert sa["rays_per_level"][0] == 160 * 90 and sa["shadow_rays"] > 0, name


def test_against_the_oracle(rta, orc):
    W, H = 96, 54
    for name in ("inside", "far_side"):
        w = _world(rta.api, 400)
        ow = _world(orc.api, 400)
        ctx = rta.Context()
        ctx.set_cull_threshold(0)
        scene = rta.Scene(ctx, rta.flatten(w))
        out, _ = _frame(rta, ctx, scene, rta.camera_desc(_camera(rta.api, W, H, VIEWS[name])), rta.F64)
        want = orc.parallel_render(_camera(orc.api, W, H, VIEWS[name]), ow)
        assert np.abs(out - want).max() <= 1e-9, name
        scene.close()
        ctx.close()


def test_wide_fields_of_view_and_small_frames(rta):
    """A tile of a 16x9 frame at 170 degrees spans more than a right angle (no cone: nothing is ruled
    out by it); frames narrower or lower than a tile, and with ragged last tiles."""
    world = _world(rta.api, 700)
    cams = {}
    for W, H, fov in [(16, 9, math.radians(170.0)), (24, 16, 3.0), (5, 3, math.pi / 3.0), (1, 1, math.pi / 3.0), (67, 41, 2.2),
                      (8, 64, math.pi / 2.0)]:
        cams["%dx%d_%g" % (W, H, fov)] = _camera(rta.api, W, H, dict(VIEWS["inside"], fov=fov))
    got = _both(rta, world, cams)
    for name in cams:
        (a, sa), (b, sb) = got[(0, name)], got[(0xffffffff, name)]
        assert sa["rays_per_level"] == sb["rays_per_level"] and sa["shadow_rays"] == sb["shadow_rays"], name
        assert np.abs(a - b).max() <= 1e-12, name


def test_lights_among_the_objects_and_objects_without_shadows(rta):
    """A light inside the cloud (spheres around and behind it, hit points on all sides); a light far
    away (every tile's hit points within a narrow cone); every third object casting no shadow, which
    takes shadow rays off the any-hit walk altogether (src/scenes/world.rs:146-158)."""
    cams = {name: _camera(rta.api, 128, 72, VIEWS[name]) for name in ("front", "inside")}
    for kw in (dict(light=(2.0, 10.0, 60.0)), dict(light=(-3000.0, 6000.0, -4000.0)), dict(no_shadow_every=3)):
        world = _world(rta.api, 900, **kw)
        got = _both(rta, world, cams)
        for name in cams:
            (a, sa), (b, sb) = got[(0, name)], got[(0xffffffff, name)]
            assert sa["rays_per_level"] == sb["rays_per_level"] and sa["shadow_rays"] == sb["shadow_rays"], (kw, name)
            assert np.abs(a - b).max() <= 1e-12, (kw, name)


def test_fast_path(rta):
    """f32: the two ways round their intersections differently where the compiler contracts them
    differently, so pixels agree to rounding, not to the bit; hits and shadows are the same rays."""
    world = _world(rta.api, 1500)
    cams = {name: _camera(rta.api, 160, 90, VIEWS[name]) for name in ("front", "inside", "far_side")}
    got = _both(rta, world, cams, rta.F32)
    for name in cams:
        (a, sa), (b, sb) = got[(0, name)], got[(0xffffffff, name)]
        close = np.abs(a - b).max(axis=2) <= 1.0 / 255.0
        assert close.mean() >= 0.999, (name, close.mean())
        assert abs(sa["shadow_rays"] - sb["shadow_rays"]) <= 20, name


@pytest.mark.parametrize("glass_every", [0, 60])
def test_lane_walk_changes_nothing(rta, glass_every):
    """Secondary rays of a scene whose packed sphere hierarchy fits LDS walk it lane by lane (and a wave per
    ray on short levels); with that switched off a wave walks the spheres in memory for its 64 rays.  The
    packed spheres hold the stored ones, so the same objects reach an exact intersection: frames equal to the
    last bit of every hit decision (colours to the few ulps the order of the framebuffer additions moves
    them), counters equal, in both precisions, with and without glass (whose container bookkeeping is a
    backward walk of its own: src/scenes/world.rs:253-285)."""
    world = _world(rta.api, 2500, glass_every=glass_every)
    flat = rta.flatten(world)
    cams = {name: _camera(rta.api, 168, 94, VIEWS[name]) for name in ("front", "inside", "far_side")}
    for precision, tol in ((rta.F64, 1e-12), (rta.F32, 2e-5)):
        got = {}
        for lane in (True, False):
            ctx = rta.Context()
            ctx.set_lane_walk(lane)
            scene = rta.Scene(ctx, flat)
            assert scene.features()["bounding_sphere_tests"] and scene.features()["lane_walk"] == lane
            for name, cam in cams.items():
                got[(lane, name)] = _frame(rta, ctx, scene, rta.camera_desc(cam), precision)
            scene.close()
            ctx.close()
        for name in cams:
            (a, sa), (b, sb) = got[(True, name)], got[(False, name)]
            assert sa["rays_per_level"] == sb["rays_per_level"] and sa["shadow_rays"] == sb["shadow_rays"], name
            assert sa["rays_per_level"][1] > 0 and (glass_every == 0 or sa["rays_per_level"][4] > 0)
            assert np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= tol, name
