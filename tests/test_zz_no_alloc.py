"""rtx_render_device_noalloc: the render call that allocates nothing (its counters, tables and ray
queues live in a file-scope device array) against rtx_render_device, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import hiprt

pytestmark = pytest.mark.gpu


def _noalloc(rta, ctx, scene, cd, depth, prec, r0, r1, ptr, nbytes, stream):
    rta._check(rta.rtx_render_device_noalloc(ctx.h, scene.h, C.byref(cd), depth, prec, r0, r1, C.c_void_p(ptr), nbytes, stream))


@pytest.mark.parametrize("prec_name", ["f64", "f32"])
def test_two_calls_in_a_row_match_the_allocating_entry_point(rta, prec_name):
    """A frame with secondary rays (queues, counters and tables all in the fixed array), then a
    smaller frame of another scene straight after it: the second call finds the first call's
    contents in the array and must clear what has to start at zero."""
    from ray_tracer_amd import scenes
    hip = hiprt.runtime()
    prec = rta.F64 if prec_name == "f64" else rta.F32
    dtype = np.float64 if prec == rta.F64 else np.float32
    tol = 1e-9 if prec == rta.F64 else 2e-5  # f64: the contract; f32: sums of a few atomics in another order
    ctx = rta.Context()
    stream = hiprt.new_stream(hip)
    def random_300(api, W, H):
        return scenes.random_scene(api, n_objects=300, hsize=W, vsize=H)

    cases = [(scenes.glass_scene, 640, 360, 5), (random_300, 320, 200, 3), (scenes.default_scene, 200, 120, 5)]
    bufs = []
    try:
        for make, W, H, depth in cases:
            w, cam = make(rta.api, W, H)
            scene = rta.Scene(ctx, rta.flatten(w))
            cd = rta.camera_desc(cam)
            nbytes = W * H * 3 * np.dtype(dtype).itemsize
            a, b = hiprt.Buffer(hip, nbytes), hiprt.Buffer(hip, nbytes)
            bufs += [a, b]
            a.fill(0xFF, stream)
            b.fill(0xFF, stream)
            rta.render_device(ctx, scene, cd, depth, prec, 0, H, a.ptr.value, nbytes, stream.value)
            _noalloc(rta, ctx, scene, cd, depth, prec, 0, H, b.ptr.value, nbytes, stream)
            want = a.read((H, W, 3), dtype, stream)
            got = b.read((H, W, 3), dtype, stream)
            assert not np.isnan(got).any()
            assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= tol
            # a row range, as a tile of a larger frame would be rendered
            _noalloc(rta, ctx, scene, cd, depth, prec, 40, 90, b.ptr.value, 50 * W * 3 * np.dtype(dtype).itemsize, stream)
            got = b.read((50, W, 3), dtype, stream)
            assert np.abs(got.astype(np.float64) - want[40:90].astype(np.float64)).max() <= tol
            ctx.synchronize()
            scene.close()
    finally:
        hip.hipStreamDestroy(stream)
        for x in bufs:
            x.free()
        ctx.close()


def test_frames_beyond_the_fixed_array_are_refused(rta):
    """An 8K frame of the glass scene at depth 7 needs more row chunks than the array has counters
    for: the call must say so and enqueue nothing."""
    from ray_tracer_amd import scenes
    hip = hiprt.runtime()
    ctx = rta.Context()
    w, cam = scenes.glass_scene(rta.api, 7680, 4320)
    scene = rta.Scene(ctx, rta.flatten(w))
    cd = rta.camera_desc(cam)
    buf = hiprt.Buffer(hip, 7680 * 4320 * 24)
    try:
        with pytest.raises(rta.RtxError, match="fixed workspace"):
            _noalloc(rta, ctx, scene, cd, 7, rta.F64, 0, 4320, buf.ptr.value, 7680 * 4320 * 24, None)
    finally:
        buf.free()
        scene.close()
        ctx.close()


def test_it_allocates_nothing_and_can_be_captured_in_a_graph(rta):
    """What the entry point is for.  (a) Device memory in use is the same before and after a run of calls
    (hipMemGetInfo; the array is resident from the first call of the process).  (b) One call recorded
    between hipStreamBeginCapture and hipStreamEndCapture -- where an allocation or a synchronisation
    would fail the capture -- and replayed twice as a graph gives the frame of rtx_render_device."""
    from ray_tracer_amd import scenes
    hip = hiprt.runtime()
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    hip.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
    hip.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipGraphExecDestroy.argtypes = [C.c_void_p]
    hip.hipGraphDestroy.argtypes = [C.c_void_p]
    W, H, depth = 480, 270, 4
    ctx = rta.Context()
    stream = hiprt.new_stream(hip)
    w, cam = scenes.glass_scene(rta.api, W, H)
    scene = rta.Scene(ctx, rta.flatten(w))
    cd = rta.camera_desc(cam)
    nbytes = W * H * 24
    a, b = hiprt.Buffer(hip, nbytes), hiprt.Buffer(hip, nbytes)
    graph, graph_exec = C.c_void_p(), C.c_void_p()
    try:
        rta.render_device(ctx, scene, cd, depth, rta.F64, 0, H, a.ptr.value, nbytes, stream.value)
        want = a.read((H, W, 3), np.float64, stream)
        _noalloc(rta, ctx, scene, cd, depth, rta.F64, 0, H, b.ptr.value, nbytes, stream)  # first use: everything is resident now
        assert hip.hipStreamSynchronize(stream) == 0
        free0, free1, total = C.c_size_t(), C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(free0), C.byref(total)) == 0
        for _ in range(5):
            _noalloc(rta, ctx, scene, cd, depth, rta.F64, 0, H, b.ptr.value, nbytes, stream)
        assert hip.hipStreamSynchronize(stream) == 0
        assert hip.hipMemGetInfo(C.byref(free1), C.byref(total)) == 0
        assert free1.value == free0.value, "device memory in use changed by %d bytes" % (free0.value - free1.value)
        # (b) as a graph
        assert hip.hipStreamBeginCapture(stream, 2) == 0  # hipStreamCaptureModeRelaxed
        try:
            _noalloc(rta, ctx, scene, cd, depth, rta.F64, 0, H, b.ptr.value, nbytes, stream)
        finally:
            assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0
        assert graph.value, "nothing was captured"
        assert hip.hipGraphInstantiate(C.byref(graph_exec), graph, None, None, 0) == 0
        for _ in range(2):
            b.fill(0xFF, stream)
            assert hip.hipGraphLaunch(graph_exec, stream) == 0
            got = b.read((H, W, 3), np.float64, stream)
            assert np.abs(got - want).max() <= 1e-9
    finally:
        if graph_exec.value:
            hip.hipGraphExecDestroy(graph_exec)
        if graph.value:
            hip.hipGraphDestroy(graph)
        hip.hipStreamDestroy(stream)
        a.free()
        b.free()
        scene.close()
        ctx.close()
