"""Cycles per phase of a color_at node in the render kernels, summed over waves, and the clock the
chip ran at during the bounce launches.  Needs a library built with the clocks:
  RTX_LIBNAME=librtx_phase.so RTX_EXTRA_FLAGS="-DRTX_PHASE_CLOCK" ./build.sh lib
then, on a GPU box:  RTX_LIB=$PWD/ray-tracer_amd/lib/librtx_phase.so python3 tools/phase_clock.py [scene] [f64|f32]
The clock reads perturb the schedule (each is a scalar memory instruction that is waited for), and a
phase's cycles include the time its wave sat waiting for the SIMD: read the shares, not the sums.
The counters of the camera kernel and of the bounce kernel live in different translation units."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
import torch
rta = load_package()
from ray_tracer_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "glass"
prec_name = sys.argv[2] if len(sys.argv) > 2 else "f64"
prec = rta.F64 if prec_name == "f64" else rta.F32
W, H = 3840, 2160
w, cam = scenes.SCENES[name](rta.api, hsize=W, vsize=H)
lib = ctypes.CDLL(os.environ["RTX_LIB"])
read_cam = lib.rtx_debug_phase if prec == rta.F64 else None
read_bounce = lib.rtx_debug_phase_bounce if prec == rta.F64 else lib.rtx_debug_phase_bounce_f32
cam_out, out = (ctypes.c_ulonglong * 16)(), (ctypes.c_ulonglong * 16)()
ctx = rta.Context(0)
scene = rta.Scene(ctx, rta.flatten(w))
cd = rta.camera_desc(cam)
tile = torch.empty((H, W, 3), dtype=torch.float64 if prec == rta.F64 else torch.float32, device="cuda")
st = {}
for _ in range(4):
    if read_cam:
        read_cam(cam_out, 1)
    read_bounce(out, 1)
    st = {}
    rta.render_device(ctx, scene, cd, 5, prec, 0, H, tile.data_ptr(), tile.numel() * tile.element_size(), None, st)
if read_cam:
    assert read_cam(cam_out, 1) == 0
assert read_bounce(out, 1) == 0
names = ["bounce: closest hit", "bounce: computations + children", "bounce: queue append", "bounce: shadow ray", "bounce: lighting",
         "bounce: whole round", "", "", "camera: closest hit", "camera: computations + children", "camera: queue append",
         "camera: shadow ray", "camera: lighting"]
print("%s %s frame %.3f ms, rays per level %s" % (name, prec_name, st["kernel_ms"], st["rays_per_level"][:6]))
for src, grp in ((out, range(0, 6)), (cam_out, range(8, 13))):
    tot = sum(src[i] for i in grp if i != 5)
    for i in grp:
        print("  %-36s %14d cycles  %5.1f %%" % (names[i], src[i], 100.0 * src[i] / max(tot, 1)))
print("  longest round of a bounce wave: %d cycles; rounds over 100 000 cycles: %d" % (out[13], out[12]))
if out[15]:
    print("  shader clock during the bounce launches: %.0f MHz (%.3f ms of launches seen by one wave each)"
          % (100.0 * out[14] / out[15], out[15] / 1e5))
