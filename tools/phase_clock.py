"""Cycles per phase of a color_at node in the render kernels, summed over waves.  Needs a library
built with the clocks:  RTX_LIBNAME=librtx_phase.so RTX_EXTRA_FLAGS="-DRTX_PHASE_CLOCK" ./build.sh lib
then, on a GPU box:  RTX_LIB=$PWD/ray-tracer_amd/lib/librtx_phase.so python3 tools/phase_clock.py [scene]
The clock reads perturb the schedule (each is a scalar memory instruction that is waited for), and a
phase's cycles include the time its wave sat waiting for the SIMD: read the shares, not the sums."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
import torch
rta = load_package()
from ray_tracer_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "glass"
W, H = 3840, 2160
w, cam = scenes.SCENES[name](rta.api, hsize=W, vsize=H)
lib = ctypes.CDLL(os.environ["RTX_LIB"])
out = (ctypes.c_ulonglong * 16)()
ctx = rta.Context(0)
scene = rta.Scene(ctx, rta.flatten(w))
cd = rta.camera_desc(cam)
tile = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
st = {}
for _ in range(3):
    lib.rtx_debug_phase(out, 1)
    st = {}
    rta.render_device(ctx, scene, cd, 5, rta.F64, 0, H, tile.data_ptr(), tile.numel() * 8, None, st)
assert lib.rtx_debug_phase(out, 1) == 0
names = ["bounce: closest hit", "bounce: computations + children", "bounce: queue append", "bounce: shadow ray", "bounce: lighting",
         "bounce: whole round", "", "", "camera: closest hit", "camera: computations + children", "camera: queue append",
         "camera: shadow ray", "camera: lighting"]
print("frame %.3f ms, rays per level %s" % (st["kernel_ms"], st["rays_per_level"][:6]))
for grp in (range(0, 6), range(8, 13)):
    tot = sum(out[i] for i in grp if i != 5)
    for i in grp:
        print("  %-36s %14d cycles  %5.1f %%" % (names[i], out[i], 100.0 * out[i] / max(tot, 1)))
