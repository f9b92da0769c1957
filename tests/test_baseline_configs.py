"""The benchmark configurations at their defining sizes: the 10 002-object scene, the default
world at 3840x2160 on the f32 path, and the 7680x4320 multi-light glass frame assembled from
the row tiles eight GPUs would render.  Each is compared with the oracle on a sample (the oracle
needs seconds per thousand rays at these sizes) and checked whole through properties."""
import ctypes as C

import numpy as np
import pytest

import hiprt

pytestmark = pytest.mark.gpu

TOL = 1e-9


def _camera_rays(orc, ocam, xs, ys):
    return np.array([orc.ray_for_pixel(ocam, int(x), int(y)) for x, y in zip(xs, ys)])


def test_ten_thousand_objects_at_full_size(orc, rta):
    """N = 10 002 (20 group spheres over 313 clusters of spheres alone): colours of 3 000 camera
    rays and 2 000 rays shot through the cloud at random against the oracle's color_at, the
    intermediate results of 600 of them bit for bit, and the 4K frame's pixels and counters."""
    from ray_tracer_amd import scenes
    W, H = 3840, 2160
    gw, gcam = scenes.random_scene(rta.api, hsize=W, vsize=H)
    ow, ocam = scenes.random_scene(orc.api, hsize=W, vsize=H)
    assert len(gw) == 10002
    rng = np.random.default_rng(2024)
    xs, ys = rng.integers(0, W, 3000), rng.integers(0, H, 3000)
    cam_rays = _camera_rays(orc, ocam, xs, ys)
    o = np.stack([rng.uniform(-45, 45, 2000), rng.uniform(0.2, 28, 2000), rng.uniform(-10, 125, 2000)], axis=1)
    target = np.stack([rng.uniform(-40, 40, 2000), rng.uniform(0.5, 25, 2000), rng.uniform(-5, 120, 2000)], axis=1)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([cam_rays, np.concatenate([o, d], axis=1)])
    g_cnt, o_cnt = [0, 0], [0, 0]
    got = rta.color_at(gw, rays, 5, counters=g_cnt)
    want = orc.color_at(ow, rays, 5, counters=o_cnt)
    assert np.abs(got - want).max() <= TOL
    assert g_cnt == o_cnt, "ray counters differ: GPU %s oracle %s" % (g_cnt, o_cnt)
    assert np.count_nonzero(want.sum(axis=1)) > 4000

    # intermediate results, bit for bit
    from test_gpu_parity import oracle_probe
    sub = rays[::8][:600]
    probes = rta.probe(gw, sub)
    for g, w_, r in zip(probes, oracle_probe(orc, ow, sub), sub):
        if w_ is None:
            assert g["object"] == -1
            continue
        t, obj, c = w_
        assert g["object"] == obj and g["t"] == t, "hit differs for ray %s" % (r,)
        assert np.array_equal(g["point"], c.point[:]) and np.array_equal(g["normalv"], c.normalv[:])
        assert np.array_equal(g["over_point"], c.over_point[:]) and np.array_equal(g["reflectv"], c.reflectv[:])
        assert bool(g["shadowed"]) == orc.is_shadowed(ow, c.over_point[:])

    # the frame itself: the sampled pixels are the colours of their rays, the counters add up
    st = {}
    img = rta.render_rows(gcam, gw, stats=st)
    assert np.abs(img[ys, xs] - want[:3000]).max() <= TOL
    assert st["rays_per_level"][0] == W * H and st["color_rays"] == sum(st["rays_per_level"])
    assert st["rays_per_level"][1] > 0 and st["rays_per_level"][5] > 0
    assert st["shadow_rays"] <= st["color_rays"]
    # every camera ray ends on the floor, the wall or an object: every pixel is a shaded hit
    assert st["shadow_rays"] >= W * H
    band = rta.render_rows(gcam, gw, row_begin=1070, row_end=1100)
    assert np.abs(band - img[1070:1100]).max() <= 1e-12


def test_ten_thousand_objects_some_of_glass(orc, rta):
    """The same cloud with every 400th object made of glass (25 of 10 002): the scene then carries the
    container bookkeeping of prepare_computations (src/scenes/world.rs:253-285), which the kernels do
    as a walk of its own for the rays that hit glass.  Colours and counters of 1 500 camera rays and
    1 500 rays aimed at the glass objects against the oracle's color_at, n1 / n2 and the rest of the
    intermediate results of 500 of them bit for bit, and the 4K frame at the sampled pixels."""
    from ray_tracer_amd import scenes
    W, H = 3840, 2160
    gw, gcam = scenes.random_scene(rta.api, hsize=W, vsize=H, glass_every=400)
    ow, ocam = scenes.random_scene(orc.api, hsize=W, vsize=H, glass_every=400)
    flat = rta.flatten(gw)
    glass = np.flatnonzero(flat["transparency"] != 0.0)
    assert len(glass) == 25 and len(gw) == 10002
    rng = np.random.default_rng(77)
    xs, ys = rng.integers(0, W, 1500), rng.integers(0, H, 1500)
    cam_rays = _camera_rays(orc, ocam, xs, ys)
    # rays towards (and, from a point inside, out of) the glass objects: centre = inverse^-1 * origin
    centres = np.array([np.linalg.inv(flat["inverse"].reshape(-1, 4, 4)[i])[:3, 3] for i in glass])
    pick = centres[rng.integers(0, len(centres), 1500)]
    o = pick + rng.normal(0.0, 1.0, (1500, 3)) * np.where(rng.uniform(size=(1500, 1)) < 0.3, 0.15, 6.0)
    target = pick + rng.normal(0.0, 0.2, (1500, 3))
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([cam_rays, np.concatenate([o, d], axis=1)])
    g_cnt, o_cnt = [0, 0], [0, 0]
    got = rta.color_at(gw, rays, 5, counters=g_cnt)
    want = orc.color_at(ow, rays, 5, counters=o_cnt)
    assert np.abs(got - want).max() <= TOL
    assert g_cnt == o_cnt, "ray counters differ: GPU %s oracle %s" % (g_cnt, o_cnt)

    from test_gpu_parity import oracle_probe
    sub = rays[1500:][::3][:500]
    probes = rta.probe(gw, sub)
    hits_on_glass = 0
    for g, w_, r in zip(probes, oracle_probe(orc, ow, sub), sub):
        if w_ is None:
            assert g["object"] == -1
            continue
        t, obj, c = w_
        assert g["object"] == obj and g["t"] == t, "hit differs for ray %s" % (r,)
        assert g["n1"] == c.n1 and g["n2"] == c.n2, "n1/n2 differ for ray %s: %s %s, oracle %s %s" % (r, g["n1"], g["n2"], c.n1, c.n2)
        assert np.array_equal(g["point"], c.point[:]) and np.array_equal(g["normalv"], c.normalv[:])
        hits_on_glass += int(obj in glass)
    assert hits_on_glass > 100

    st = {}
    img = rta.render_rows(gcam, gw, stats=st)
    assert np.abs(img[ys, xs] - want[:1500]).max() <= TOL
    assert st["rays_per_level"][0] == W * H and st["rays_per_level"][5] > 0
    # the fast path: at least 99.9 % of the sampled pixels within one 8-bit level
    img32 = rta.render_rows(gcam, gw, precision=rta.F32)
    q_got = orc.quantize_u8(img32[ys, xs].astype(np.float64)).astype(np.int32)
    q_want = orc.quantize_u8(want[:1500]).astype(np.int32)
    assert np.all(np.abs(q_got - q_want) <= 1, axis=1).mean() >= 0.995


def test_default_world_4k_on_the_f32_path(orc, rta):
    """BASELINE config 2 in f32: after the reference's 8-bit conversion (src/scenes/canvas.rs:84-86)
    at least 99.9 % of the pixels are within one level of the f64 oracle, on 40 000 sampled
    pixels, half of them where the spheres are."""
    from ray_tracer_amd import scenes
    W, H = 3840, 2160
    gw, gcam = scenes.default_scene(rta.api, W, H)
    ow, ocam = scenes.default_scene(orc.api, W, H)
    st = {}
    img = rta.render_rows(gcam, gw, precision=rta.F32, stats=st)
    assert st["color_rays"] == W * H
    rng = np.random.default_rng(5)
    n = 40000
    xs, ys = rng.integers(0, W, n), rng.integers(0, H, n)
    xs[: n // 2] = rng.integers(1500, 2340, n // 2)
    ys[: n // 2] = rng.integers(1300, 2000, n // 2)
    want = orc.color_at(ow, _camera_rays(orc, ocam, xs, ys))
    q_got = orc.quantize_u8(img[ys, xs].astype(np.float64)).astype(np.int32)
    q_want = orc.quantize_u8(want).astype(np.int32)
    close = np.all(np.abs(q_got - q_want) <= 1, axis=1)
    assert close.mean() >= 0.999, "only %.4f %% of the pixels within 1/255" % (100.0 * close.mean())
    assert np.count_nonzero(want.sum(axis=1)) > 5000
    # the frame's lit pixels are those of the f64 frame, give or take the silhouettes
    img64 = rta.render_rows(gcam, gw)
    differ = np.count_nonzero((img.sum(axis=2) > 0) != (img64.sum(axis=2) > 0))
    assert differ <= 0.0005 * W * H


def test_8k_multi_light_frame_from_eight_row_tiles(orc, rta):
    """BASELINE config 5 up to the gather: the 7680x4320 glass frame with four lights in the World
    (only lights[0] shades, src/scenes/world.rs:123,147), rendered once whole and once as the
    eight row tiles of whole 10-row bands that eight ranks would render, into one buffer."""
    from ray_tracer_amd import scenes, sharding
    W, H = 7680, 4320
    hip = hiprt.runtime()
    gw, gcam = scenes.glass_scene(rta.api, W, H, n_lights=4)
    ow, ocam = scenes.glass_scene(orc.api, W, H, n_lights=4)
    flat = rta.flatten(gw)
    assert flat["light_position"].size == 12
    ctx = rta.Context()
    scene = rta.Scene(ctx, flat)
    cd = rta.camera_desc(gcam)
    nbytes = W * H * 24
    whole, tiled = hiprt.Buffer(hip, nbytes), hiprt.Buffer(hip, nbytes)
    stream = hiprt.new_stream(hip)
    try:
        st = {}
        rta.render_device(ctx, scene, cd, 5, rta.F64, 0, H, whole.ptr.value, nbytes, stream.value, st)
        tiles = sharding.row_tiles(H, 8)
        assert tiles[0][0] == 0 and tiles[-1][1] == H and all(a % 10 == 0 for a, _ in tiles)
        tile_rays = 0
        for r0, r1 in tiles:
            ts = {}
            rta.render_device(ctx, scene, cd, 5, rta.F64, r0, r1, tiled.ptr.value + r0 * W * 24, (r1 - r0) * W * 24, stream.value, ts)
            tile_rays += ts["color_rays"] + ts["shadow_rays"]
        assert tile_rays == st["color_rays"] + st["shadow_rays"]
        a = whole.read((H, W, 3), np.float64, stream)
        b = tiled.read((H, W, 3), np.float64, stream)
        assert np.abs(a - b).max() <= 1e-12
        rng = np.random.default_rng(8)
        xs, ys = rng.integers(0, W, 1500), rng.integers(0, H, 1500)
        want = orc.color_at(ow, _camera_rays(orc, ocam, xs, ys))
        assert np.abs(a[ys, xs] - want).max() <= TOL
        assert st["rays_per_level"][0] == W * H and st["rays_per_level"][5] > 0
        # ... and as it ships (rtx_multi_render, bench.py): the 10-row bands dealt out in turn to eight parts
        # (band b to part b mod 8, src/scenes/camera.rs:177), each part's bands back to back in its own buffer
        # (rtx_render_part_device), values and RGBA8; put back in place they are the whole frame
        part_rays = 0
        bands = np.arange((H + 9) // 10)
        quant = orc.quantize_u8(a)
        for k in range(8):
            rows = rta.rtx_part_rows(H, 10, 8, k)
            image_rows = np.concatenate([np.arange(b * 10, min(b * 10 + 10, H)) for b in bands[k::8]])
            assert rows == len(image_rows)
            ps = {}
            rta.render_part_device(ctx, scene, cd, 5, rta.F64, 10, 8, k, 0, tiled.ptr.value, rows * W * 24, stream.value, ps)
            part_rays += ps["color_rays"] + ps["shadow_rays"]
            got = tiled.read((rows, W, 3), np.float64, stream)
            assert np.abs(got - a[image_rows]).max() <= 1e-12, "part %d of 8" % k
            rta.render_part_device(ctx, scene, cd, 5, rta.F64, 10, 8, k, rta.PIXELS_RGBA8, tiled.ptr.value, rows * W * 4, stream.value)
            got8 = tiled.read((rows, W, 4), np.uint8, stream)
            diff = np.abs(got8[..., :3].astype(np.int16) - quant[image_rows].astype(np.int16))
            assert diff.max() <= 1 and (diff != 0).mean() < 1e-4 and (got8[..., 3] == 255).all()
        assert part_rays == st["color_rays"] + st["shadow_rays"]
    finally:
        hip.hipStreamDestroy(stream)
        whole.free()
        tiled.free()
        scene.close()
        ctx.close()
