"""Scenes walked object by object keep the secondary rays of a screen region, and of one kind (reflection,
refraction), in sub-queues of their own and take the blocks of all sub-queues from one list.  The rays are the
reference's rays whatever queue they wait in (World::reflected_color / refracted_color,
src/scenes/world.rs:160-190): frames of awkward sizes -- widths that are no multiple of a camera workgroup,
fewer rows than rows of regions, one column of regions, more regions than pixels -- are compared with the
oracle's color_at on a sample of pixels, and sized both ways (worst case on the host; level by level from
measured lengths, where the sub-queues of a queue differ in capacity) they hold the same rays."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NEVER = (1 << 64) - 1


def _render(rta, ctx, scene, cd, W, H, depth, precision=None):
    precision = rta.F64 if precision is None else precision
    out = np.empty((H, W, 3), dtype=np.float64 if precision == rta.F64 else np.float32)
    st = rta.Stats()
    rta._check(rta.rtx_render(ctx.h, scene.h, cd, depth, precision, 0, H, out.ctypes.data, out.nbytes, None, None, C.byref(st)))
    return out, st.as_dict()


@pytest.mark.parametrize("size", [(1000, 701), (70, 3), (64, 1500), (37, 41), (2050, 130)])
def test_frames_of_awkward_sizes(rta, orc, size):
    from ray_tracer_amd import scenes
    W, H = size
    w, cam = scenes.glass_scene(rta.api, W, H)
    ow, ocam = scenes.glass_scene(orc.api, W, H)
    flat = rta.flatten(w)
    cd = rta.camera_desc(cam)
    frames, stats = [], []
    for above in (NEVER, 0):
        ctx = rta.Context()
        ctx.set_measured_queues(above)
        scene = rta.Scene(ctx, flat)
        img, st = _render(rta, ctx, scene, cd, W, H, 5)
        frames.append(img)
        stats.append(st)
        scene.close()
        ctx.close()
    assert np.abs(frames[0] - frames[1]).max() <= 1e-12
    assert stats[0]["rays_per_level"] == stats[1]["rays_per_level"]
    assert stats[0]["shadow_rays"] == stats[1]["shadow_rays"]
    rng = np.random.default_rng(W * 7919 + H)
    n = min(250, W * H)
    xs, ys = rng.integers(0, W, n), rng.integers(0, H, n)
    rays = np.array([orc.ray_for_pixel(ocam, int(x), int(y)) for x, y in zip(xs, ys)])
    want = orc.color_at(ow, rays)
    assert np.abs(frames[0][ys, xs] - want).max() <= 1e-9
    assert np.abs(frames[1][ys, xs] - want).max() <= 1e-9


def test_reflection_only_and_refraction_only_scenes(rta, orc):
    """Scenes whose rays spawn one kind of child have one sub-queue per region."""
    from ray_tracer_amd import scenes
    W, H = 333, 190
    for api_name in ("mirror", "clear"):
        worlds = []
        for api in (rta.api, orc.api):
            w = api.world()
            floor = api.plane()
            ball = api.sphere().translate(0.0, 1.0, 0.0)
            if api_name == "mirror":
                floor.set_reflective(0.4)
                ball.set_reflective(0.8).set_color(0.2, 0.3, 0.9)
            else:
                ball.set_transparency(0.9).set_refractive_index(1.5).set_color(0.1, 0.1, 0.1)
                w.add_object(api.sphere().translate(0.0, 1.0, 0.0).scale(0.4, 0.4, 0.4).set_transparency(1.0).set_refractive_index(1.0))
            w.add_object(floor)
            w.add_object(ball)
            w.add_object(api.cube().translate(2.0, 0.5, 1.0).scale(0.5, 0.5, 0.5).set_color(0.9, 0.4, 0.1))
            w.add_light((-10.0, 10.0, -10.0))
            cam = scenes._camera(api, W, H, 1.0, (0.0, 2.0, -6.0), (0.0, 1.0, 0.0))
            worlds.append((w, cam))
        (w, cam), (ow, ocam) = worlds
        ctx = rta.Context()
        scene = rta.Scene(ctx, rta.flatten(w))
        feats = scene.features()
        assert feats["reflective"] != feats["transparent"]
        img, st = _render(rta, ctx, scene, rta.camera_desc(cam), W, H, 5)
        assert sum(st["rays_per_level"][1:]) > 0
        rng = np.random.default_rng(11)
        xs, ys = rng.integers(0, W, 300), rng.integers(0, H, 300)
        rays = np.array([orc.ray_for_pixel(ocam, int(x), int(y)) for x, y in zip(xs, ys)])
        assert np.abs(img[ys, xs] - orc.color_at(ow, rays)).max() <= 1e-9
        scene.close()
        ctx.close()
