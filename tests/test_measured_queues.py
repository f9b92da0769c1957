"""Frames whose secondary-ray queues are sized level by level from the lengths read back
(rtx_context_set_measured_queues) against the same frames with every queue sized for the worst
case on the host: the rays are the same rays, so counters are equal and pixels agree to the
order of the sums (World::color_at recursion, src/scenes/world.rs:121-190)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NEVER = (1 << 64) - 1


def _render(rta, ctx, scene, cd, W, H, depth, precision=None, rows=None):
    precision = rta.F64 if precision is None else precision
    r0, r1 = rows or (0, H)
    out = np.empty((r1 - r0, W, 3), dtype=np.float64 if precision == rta.F64 else np.float32)
    st = rta.Stats()
    rta._check(rta.rtx_render(ctx.h, scene.h, cd, depth, precision, r0, r1, out.ctypes.data, out.nbytes, None, None, C.byref(st)))
    return out, st.as_dict()


@pytest.mark.parametrize("depth", [2, 5, 7])
def test_measured_frames_equal_worst_case_frames(rta, depth):
    from ray_tracer_amd import scenes
    W, H = 640, 360
    w, cam = scenes.glass_scene(rta.api, W, H)
    flat = rta.flatten(w)
    cd = rta.camera_desc(cam)
    frames, stats, held = [], [], []
    for above in (NEVER, 0):
        ctx = rta.Context()
        ctx.set_measured_queues(above)
        scene = rta.Scene(ctx, flat)
        img, st = _render(rta, ctx, scene, cd, W, H, depth)
        again, st2 = _render(rta, ctx, scene, cd, W, H, depth)
        assert np.abs(img - again).max() <= 1e-12
        assert st["rays_per_level"] == st2["rays_per_level"]
        frames.append(img)
        stats.append(st)
        held.append(ctx.queue_bytes())
        scene.close()
        ctx.close()
    assert np.abs(frames[0] - frames[1]).max() <= 1e-12
    assert stats[0]["rays_per_level"] == stats[1]["rays_per_level"]
    assert stats[0]["shadow_rays"] == stats[1]["shadow_rays"]
    assert stats[0]["n_launches"] == stats[1]["n_launches"] == depth + 1
    # the worst case doubles per level, the rays of this scene do not
    if depth >= 5:
        assert held[1] * 4 < held[0], held


def test_measured_frames_in_row_chunks_and_ray_batches(rta, orc):
    """A workspace limit that forces several chunks (the first level of each is still planned for
    the worst case of its primary rays), the f32 path, and rtx_trace_rays, which always waits."""
    from ray_tracer_amd import scenes
    W, H = 640, 360
    w, cam = scenes.glass_scene(rta.api, W, H)
    flat = rta.flatten(w)
    cd = rta.camera_desc(cam)
    plain = rta.Context()
    plain.set_measured_queues(NEVER)
    scene_p = rta.Scene(plain, flat)
    measured = rta.Context()
    measured.set_measured_queues(0)
    measured.set_workspace_limit(24 << 20)
    scene_m = rta.Scene(measured, flat)
    want, st_w = _render(rta, plain, scene_p, cd, W, H, 5)
    got, st_g = _render(rta, measured, scene_m, cd, W, H, 5)
    assert st_g["n_chunks"] > 1
    assert np.abs(want - got).max() <= 1e-12
    assert st_w["rays_per_level"] == st_g["rays_per_level"]
    want32, _ = _render(rta, plain, scene_p, cd, W, H, 5, rta.F32, rows=(100, 200))
    got32, _ = _render(rta, measured, scene_m, cd, W, H, 5, rta.F32, rows=(100, 200))
    assert np.abs(want32 - got32).max() <= 1e-5
    # a batch of rays through the measured path against the oracle's color_at
    ow, ocam = scenes.glass_scene(orc.api, W, H)
    rng = np.random.default_rng(5)
    xs, ys = rng.integers(0, W, 300), rng.integers(0, H, 300)
    rays = np.array([orc.ray_for_pixel(ocam, int(x), int(y)) for x, y in zip(xs, ys)])
    out = np.empty((len(rays), 3))
    st = rta.Stats()
    r = np.ascontiguousarray(rays)
    dp = C.POINTER(C.c_double)
    rta._check(rta.rtx_trace_rays(measured.h, scene_m.h, r.ctypes.data_as(dp), len(r), 5, rta.F64, out.ctypes.data_as(dp), C.byref(st)))
    assert np.abs(out - orc.color_at(ow, rays)).max() <= 1e-9
    for s in (scene_p, scene_m):
        s.close()
    plain.close()
    measured.close()


def test_measured_frames_of_a_scene_with_the_sphere_hierarchy(rta):
    """Hundreds of objects, some of glass: the bounce levels run the kernel that keeps the packed sphere hierarchy
    in LDS and hands a sub-queue out a wave at a time, one lane or 64 lanes per ray by the level's length; the host
    repeats that choice when it sizes the next level.  Same frame, same counters, a fraction of the memory."""
    from ray_tracer_amd import scenes
    W, H = 640, 360
    w, cam = scenes.random_scene(rta.api, n_objects=600, hsize=W, vsize=H, glass_every=20)
    flat = rta.flatten(w)
    cd = rta.camera_desc(cam)
    frames, stats, held = [], [], []
    for above in (NEVER, 0):
        ctx = rta.Context()
        ctx.set_measured_queues(above)
        scene = rta.Scene(ctx, flat)
        feats = scene.features()
        assert feats["bounding_sphere_tests"] and feats["lane_walk"] and feats["transparent"], feats
        img, st = _render(rta, ctx, scene, cd, W, H, 5)
        again, st2 = _render(rta, ctx, scene, cd, W, H, 5)
        assert np.abs(img - again).max() <= 1e-12
        assert st["rays_per_level"] == st2["rays_per_level"]
        frames.append(img)
        stats.append(st)
        held.append(ctx.queue_bytes())
        scene.close()
        ctx.close()
    assert np.abs(frames[0] - frames[1]).max() <= 1e-12
    assert stats[0]["rays_per_level"] == stats[1]["rays_per_level"]
    assert stats[0]["shadow_rays"] == stats[1]["shadow_rays"]
    assert stats[0]["rays_per_level"][5] > 0
    assert held[1] * 4 < held[0], held
