"""Where does the f32 fast path differ from the f64 path?  Run on a GPU box:
python3 tools/f32_diff.py [scene] [hsize] [vsize]
Prints the share of pixels whose 8-bit value differs by more than 1, split by the object the
primary ray hits, and a coarse map of where they are."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
rta = load_package()
from ray_tracer_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "glass"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 320
H = int(sys.argv[3]) if len(sys.argv) > 3 else 180
kw = dict(hsize=W, vsize=H)
if name == "random":
    kw["n_objects"] = int(os.environ.get("OBJECTS", "300"))
w, cam = scenes.SCENES[name](rta.api, **kw)
a = rta.render_rows(cam, w, precision=rta.F64)
b = rta.render_rows(cam, w, precision=rta.F32)
if os.environ.get("SAVE"):   # keep this library's f64 frame ...
    np.save(os.environ["SAVE"], a)
if os.environ.get("REF"):    # ... and compare another library's f64 frame against it
    b = a
    a = np.load(os.environ["REF"])
q = lambda v: np.clip(np.nan_to_num(v * 255.999), 0, 255).astype(np.uint8).astype(int)
bad = np.abs(q(a) - q(b)).max(axis=2) > 1
print("%s %dx%d: %.4f%% of pixels differ by more than 1/255 (%d)" % (name, W, H, 100.0 * bad.mean(), bad.sum()))
zoom = [int(v) for v in os.environ.get("ZOOM", "0,0,%d,%d" % (W, H)).split(",")]
if os.environ.get("SKIP_PROBE"):
    obj = np.zeros((H, W), int)
    pr = {"t": np.zeros(H * W)}
else:
    rays = np.array([rta.ray_for_pixel(cam, x, y) for y in range(H) for x in range(W)]).reshape(-1, 6)
    pr = rta.probe(w, rays)
    obj = pr["object"].reshape(H, W)
for o in np.unique(obj):
    m = obj == o
    print("  primary hit object %3d: %7d pixels, %6d bad (%.3f%%)" % (o, m.sum(), (bad & m).sum(), 100.0 * (bad & m).sum() / m.sum()))
d = np.abs(a - b).max(axis=2)
print("  max abs colour diff %.4f, mean over bad pixels %.4f" % (d.max(), d[bad].mean() if bad.any() else 0.0))
gy, gx = 50, 100
for j in range(gy):
    row = ""
    for i in range(gx):
        zx0, zy0, zx1, zy1 = zoom
        blk = bad[zy0 + j * (zy1 - zy0) // gy:zy0 + (j + 1) * (zy1 - zy0) // gy, zx0 + i * (zx1 - zx0) // gx:zx0 + (i + 1) * (zx1 - zx0) // gx]
        f = blk.mean() if blk.size else 0
        row += " .:*#"[min(4, int(f * 20 + (f > 0)))]
    print(row)
for depth in (0, 1, 2, 5):
    a_d = rta.render_rows(cam, w, depth=depth, precision=rta.F64)
    b_d = rta.render_rows(cam, w, depth=depth, precision=rta.F32)
    print("depth %d: %d bad" % (depth, (np.abs(q(a_d) - q(b_d)).max(axis=2) > 1).sum()))
ys, xs = np.nonzero(bad)
t = pr["t"].reshape(H, W)
for y, x in list(zip(ys, xs))[:: max(1, len(ys) // 40)]:
    print("  (%3d,%3d) obj %d t %.4f  f64 %s  f32 %s" % (x, y, obj[y, x], t[y, x], np.round(a[y, x], 3), np.round(b[y, x], 3)))
