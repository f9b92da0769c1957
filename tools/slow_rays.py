"""Which rays make a bounce wave of the 10 002-object frame slow.  Needs the phase-clock build
(see phase_clock.py).  A frame is rendered, the library keeps the 64 rays of the first wave whose
round took over `threshold` cycles, and each of them is then traced on its own (depth 0: the ray
and its shadow ray) in both precisions, timed by the library's events.
  RTX_LIB=$PWD/ray-tracer_amd/lib/librtx_phase.so python3 tools/slow_rays.py [f32|f64] [threshold cycles]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
import torch, numpy as np
rta = load_package()
from ray_tracer_amd import scenes
prec_name = sys.argv[1] if len(sys.argv) > 1 else "f32"
threshold = int(sys.argv[2]) if len(sys.argv) > 2 else 3000000
prec = rta.F32 if prec_name == "f32" else rta.F64
fetch = getattr(rta.lib, "rtx_debug_slow_rays_" + prec_name)
fetch.argtypes = [C.POINTER(C.c_double), C.c_ulonglong]
W, H = 3840, 2160
w, cam = scenes.SCENES["random"](rta.api, hsize=W, vsize=H)
ctx = rta.Context(0); scene = rta.Scene(ctx, rta.flatten(w)); cd = rta.camera_desc(cam)
tile = torch.empty((H, W, 3), dtype=torch.float32 if prec == rta.F32 else torch.float64, device="cuda")
out = np.zeros(64 * 8)
assert fetch(out.ctypes.data_as(C.POINTER(C.c_double)), threshold) == 0  # arm
rta.render_device(ctx, scene, cd, 5, prec, 0, H, tile.data_ptr(), tile.numel() * tile.element_size(), None, {})
assert fetch(out.ctypes.data_as(C.POINTER(C.c_double)), threshold) == 0
rays = out.reshape(64, 8)
np.set_printoptions(linewidth=200, precision=9)
print("round of %d cycles; lane  f32 ms   f64 ms   origin, direction" % rays[0, 7])
res = np.empty(3)
for i in range(64):
    r = np.ascontiguousarray(rays[i, :6])
    ms = []
    for p in (rta.F32, rta.F64):
        best = 1e9
        for _ in range(3):
            st = rta.Stats()
            rta._check(rta.rtx_trace_rays(ctx.h, scene.h, r.ctypes.data_as(C.POINTER(C.c_double)), 1, 0, p,
                                          res.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st)))
            best = min(best, st.kernel_ms)
        ms.append(best)
    print("%3d  %7.3f  %7.3f  %s  colour %s" % (i, ms[0], ms[1], r, res))
