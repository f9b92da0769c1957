"""Frames rendered in parts (rtx_render_part_device): the bands of 10 rows dealt out in turn to
several GPUs, as values or as 8-bit pixels converted on the device, reassembled here on the host
and compared with the whole frame."""
import ctypes as C

import numpy as np
import pytest

import hiprt

pytestmark = pytest.mark.gpu


def part_image_rows(vsize, band, n_parts, part):
    """Image rows of one part, in the order its buffer holds them."""
    rows = []
    for b in range(part, (vsize + band - 1) // band, n_parts):
        rows.extend(range(b * band, min((b + 1) * band, vsize)))
    return rows


@pytest.mark.parametrize("scene_name,W,H,n_parts", [("default", 250, 197, 3), ("glass", 320, 183, 8), ("random", 200, 120, 5)])
def test_parts_reassemble_to_the_whole_frame(rta, orc, scene_name, W, H, n_parts):
    from ray_tracer_amd import scenes
    hip = hiprt.runtime()
    make = {"default": scenes.default_scene, "glass": scenes.glass_scene,
            "random": lambda api, w, h: scenes.random_scene(api, n_objects=300, hsize=w, vsize=h)}[scene_name]
    w, cam = make(rta.api, W, H)
    ctx = rta.Context()
    scene = rta.Scene(ctx, rta.flatten(w))
    cd = rta.camera_desc(cam)
    whole = np.empty((H, W, 3))
    st_whole = rta.Stats()
    rta._check(rta.rtx_render(ctx.h, scene.h, cd, 5, rta.F64, 0, H, whole.ctypes.data, whole.nbytes, None, None, st_whole))
    buf = hiprt.Buffer(hip, W * H * 24)
    stream = hiprt.new_stream(hip)
    try:
        for fmt, dtype, chans in [(0, np.float64, 3), (rta.PIXELS_RGBA8, np.uint8, 4), (rta.PIXELS_RGB8, np.uint8, 3),
                                  (rta.PIXELS_PPM8, np.uint8, 3)]:
            frame = np.zeros((H, W, chans), dtype=dtype)
            rays = 0
            for part in range(n_parts):
                rows = part_image_rows(H, 10, n_parts, part)
                nbytes = len(rows) * W * chans * np.dtype(dtype).itemsize
                buf.fill(0xEE, stream)
                st = {}
                rta.render_part_device(ctx, scene, cd, 5, rta.F64, 10, n_parts, part, fmt, buf.ptr.value, nbytes, stream.value, st)
                rays += st["color_rays"] + st["shadow_rays"]
                if rows:
                    frame[rows] = buf.read((len(rows), W, chans), dtype, stream)
            assert rays == st_whole.color_rays + st_whole.shadow_rays
            if fmt == 0:
                assert np.abs(frame - whole).max() <= 1e-12
            elif fmt == rta.PIXELS_PPM8:
                # pixels whose value sits within rounding of a step of the conversion may differ by one
                # level in frames whose pixels are sums of atomics; elsewhere the bytes are equal
                want = orc.quantize_ppm(whole)
                diff = np.abs(frame.astype(int) - want.astype(int))
                assert diff.max() <= (0 if scene_name == "default" else 1) and (diff != 0).mean() < 1e-3
            else:
                want = orc.quantize_u8(whole)
                diff = np.abs(frame[..., :3].astype(int) - want.astype(int))
                assert diff.max() <= (0 if scene_name == "default" else 1) and (diff != 0).mean() < 1e-3
                if chans == 4:
                    assert (frame[..., 3] == 255).all()
    finally:
        hip.hipStreamDestroy(stream)
        buf.free()
        scene.close()
        ctx.close()


def test_8bit_frame_without_secondary_rays_is_one_launch(rta, orc):
    """The default world has no reflective or transparent object: its RGBA8 frame is written by the
    kernel that shades the pixels, the reference's conversion (src/scenes/canvas.rs:94-104) included."""
    from ray_tracer_amd import scenes
    hip = hiprt.runtime()
    W, H = 640, 360
    w, cam = scenes.default_scene(rta.api, W, H)
    ctx = rta.Context()
    scene = rta.Scene(ctx, rta.flatten(w))
    cd = rta.camera_desc(cam)
    buf = hiprt.Buffer(hip, W * H * 4)
    try:
        st = {}
        rta.render_part_device(ctx, scene, cd, 5, rta.F64, 10, 1, 0, rta.PIXELS_RGBA8, buf.ptr.value, W * H * 4, None, st)
        assert st["n_launches"] == 1
        got = buf.read((H, W, 4), np.uint8)
        whole = np.empty((H, W, 3))
        rta._check(rta.rtx_render(ctx.h, scene.h, cd, 5, rta.F64, 0, H, whole.ctypes.data, whole.nbytes, None, None, None))
        assert np.array_equal(got[..., :3], orc.quantize_u8(whole)) and (got[..., 3] == 255).all()
    finally:
        buf.free()
        scene.close()
        ctx.close()


@pytest.mark.parametrize("scene_name,W,H,ids", [("default", 250, 197, [0, 0, 0]), ("glass", 320, 183, [0, 0]),
                                                ("glass", 200, 95, [0] * 8)])
def test_several_gpus_behind_the_c_abi(rta, orc, scene_name, W, H, ids):
    """rtx_multi_*: the single-process path over n GPUs, rehearsed here with the one GPU of the box
    named n times (n contexts, n scene uploads, n streams, bands dealt out in turn, strided copies
    into the frame): values into host memory, RGBA8 into device memory."""
    from ray_tracer_amd import scenes
    hip = hiprt.runtime()
    make = {"default": scenes.default_scene, "glass": scenes.glass_scene}[scene_name]
    w, cam = make(rta.api, W, H)
    flat = rta.flatten(w)
    cd = rta.camera_desc(cam)
    ctx = rta.Context()
    scene = rta.Scene(ctx, flat)
    whole = np.empty((H, W, 3))
    st_whole = rta.Stats()
    rta._check(rta.rtx_render(ctx.h, scene.h, cd, 5, rta.F64, 0, H, whole.ctypes.data, whole.nbytes, None, None, st_whole))
    multi = rta.Multi(ids)
    assert rta.rtx_multi_device_count(multi.h) == len(ids)
    ms = multi.upload(flat)
    buf = hiprt.Buffer(hip, W * H * 4)
    try:
        got = np.zeros((H, W, 3))
        st = {}
        multi.render(ms, cd, 5, rta.F64, 0, got.ctypes.data, got.nbytes, flags=rta.MULTI_COUNT_RAYS, stats=st)
        assert np.abs(got - whole).max() <= 1e-12
        assert st["color_rays"] + st["shadow_rays"] == st_whole.color_rays + st_whole.shadow_rays
        assert st["n_devices"] == len(ids) and st["render_ms"] > 0
        buf.fill(0x11)
        st = {}
        multi.render(ms, cd, 5, rta.F64, rta.PIXELS_RGBA8, buf.ptr.value, W * H * 4, on_device=True, stats=st)
        rgba = buf.read((H, W, 4), np.uint8)
        diff = np.abs(rgba[..., :3].astype(int) - orc.quantize_u8(whole).astype(int))
        assert diff.max() <= (0 if scene_name == "default" else 1) and (diff != 0).mean() < 1e-3
        assert (rgba[..., 3] == 255).all()
        part0 = rta.rtx_part_rows(H, 10, len(ids), 0)
        assert st["gather_bytes"] == (H - part0) * W * 4
        with pytest.raises(rta.MultiError, match="too small"):
            multi.render(ms, cd, 5, rta.F64, 0, got.ctypes.data, got.nbytes - 8)
    finally:
        buf.free()
        rta.rtx_multi_scene_destroy(ms)
        multi.close()
        scene.close()
        ctx.close()


@pytest.mark.gpu
def test_two_different_gpus_behind_the_c_abi(rta, orc):
    """rtx_multi_* on two physical GPUs (peer copies over xGMI, or through host memory where the GPUs
    cannot reach each other): the frame equals the one GPU's, as values and as RGBA8.  Skipped on a
    box with one GPU: there the same code runs with one GPU named several times (above)."""
    hip = hiprt.runtime()
    count = C.c_int(0)
    assert hip.hipGetDeviceCount(C.byref(count)) == 0
    if count.value < 2:
        pytest.skip("needs two GPUs")
    from ray_tracer_amd import scenes
    W, H = 640, 367
    w, cam = scenes.glass_scene(rta.api, W, H)
    flat = rta.flatten(w)
    cd = rta.camera_desc(cam)
    ctx = rta.Context(0)
    scene = rta.Scene(ctx, flat)
    whole = np.empty((H, W, 3))
    rta._check(rta.rtx_render(ctx.h, scene.h, cd, 5, rta.F64, 0, H, whole.ctypes.data, whole.nbytes, None, None, None))
    multi = rta.Multi([0, 1])
    ms = multi.upload(flat)
    buf = hiprt.Buffer(hip, W * H * 4)
    try:
        got = np.zeros((H, W, 3))
        multi.render(ms, cd, 5, rta.F64, 0, got.ctypes.data, got.nbytes)
        assert np.abs(got - whole).max() <= 1e-12
        buf.fill(0x11)
        multi.render(ms, cd, 5, rta.F64, rta.PIXELS_RGBA8, buf.ptr.value, W * H * 4, on_device=True)
        rgba = buf.read((H, W, 4), np.uint8)
        diff = np.abs(rgba[..., :3].astype(int) - orc.quantize_u8(whole).astype(int))
        assert diff.max() <= 1 and (diff != 0).mean() < 1e-3
    finally:
        buf.free()
        rta.rtx_multi_scene_destroy(ms)
        multi.close()
        scene.close()
        ctx.close()
