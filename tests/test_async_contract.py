"""What a caller of rtx_render_device may rely on (include/rtx.h): with no stream of its own the
call returns a finished frame; with one, everything is ordered on that stream; a secondary-ray
queue that overflows is reported, not swallowed."""
import numpy as np
import pytest

import hiprt

pytestmark = pytest.mark.gpu


def test_without_a_stream_the_frame_is_complete_on_return(rta):
    """The buffer is read on another non-blocking stream with no device-wide wait in between:
    only a call that has waited for its own kernels makes that safe."""
    from ray_tracer_amd import scenes
    hip = hiprt.runtime()
    ctx = rta.Context()
    w, cam = scenes.glass_scene(rta.api, 640, 360)
    scene = rta.Scene(ctx, rta.flatten(w))
    cd = rta.camera_desc(cam)
    want = np.empty((360, 640, 3))
    rta._check(rta.rtx_render(ctx.h, scene.h, cd, 5, rta.F64, 0, 360, want.ctypes.data, want.nbytes, None, None, None))
    buf = hiprt.Buffer(hip, want.nbytes)
    other = hiprt.new_stream(hip)
    try:
        for _ in range(4):
            buf.fill(0xFF, other)
            assert hip.hipStreamSynchronize(other) == 0
            rta.render_device(ctx, scene, cd, 5, rta.F64, 0, 360, buf.ptr.value, want.nbytes, None)
            got = buf.read(want.shape, np.float64, other)
            assert not np.isnan(got).any()
            assert np.abs(got - want).max() <= 1e-12
    finally:
        hip.hipStreamDestroy(other)
        buf.free()
        scene.close()
        ctx.close()


def test_a_queue_overflow_is_reported(rta):
    """Queues a sixty-fourth of their planned size cannot hold the glass scene's rays.  The call
    that waits reports it itself; after a call that does not wait, rtx_context_synchronize (or the
    next render call) does.  The report is made once, and frames rendered afterwards with the
    normal sizes are right."""
    from ray_tracer_amd import scenes
    hip = hiprt.runtime()
    ctx = rta.Context()
    w, cam = scenes.glass_scene(rta.api, 640, 360)
    scene = rta.Scene(ctx, rta.flatten(w))
    cd = rta.camera_desc(cam)
    want = np.empty((360, 640, 3))
    rta._check(rta.rtx_render(ctx.h, scene.h, cd, 5, rta.F64, 0, 360, want.ctypes.data, want.nbytes, None, None, None))
    buf = hiprt.Buffer(hip, want.nbytes)
    stream = hiprt.new_stream(hip)
    try:
        ctx.set_debug_queue_divisor(64)
        out = np.empty_like(want)
        with pytest.raises(rta.RtxError, match="overflow"):
            st = {}
            rta.render_device(ctx, scene, cd, 5, rta.F64, 0, 360, buf.ptr.value, want.nbytes, None, st)
        ctx.synchronize()  # reported once
        rta.render_device(ctx, scene, cd, 5, rta.F64, 0, 360, buf.ptr.value, want.nbytes, stream.value)  # returns at once
        assert hip.hipStreamSynchronize(stream) == 0
        with pytest.raises(rta.RtxError, match="overflow"):
            ctx.synchronize()
        rta.render_device(ctx, scene, cd, 5, rta.F64, 0, 360, buf.ptr.value, want.nbytes, stream.value)
        assert hip.hipStreamSynchronize(stream) == 0
        ctx.set_debug_queue_divisor(1)
        with pytest.raises(rta.RtxError, match="overflow"):  # the next render call reports it too
            rta._check(rta.rtx_render(ctx.h, scene.h, cd, 5, rta.F64, 0, 360, out.ctypes.data, out.nbytes, None, None, None))
        rta._check(rta.rtx_render(ctx.h, scene.h, cd, 5, rta.F64, 0, 360, out.ctypes.data, out.nbytes, None, None, None))
        assert np.abs(out - want).max() <= 1e-12
        ctx.synchronize()
    finally:
        hip.hipStreamDestroy(stream)
        buf.free()
        scene.close()
        ctx.close()
