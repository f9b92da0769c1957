import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from __graft_entry__ import load_package
rta = load_package()
import test_gpu_parity as T
w, cam = T.shapes_scene(rta.api, kind=0)
a = rta.render_rows(cam, w, precision=rta.F64)
b = rta.render_rows(cam, w, precision=rta.F32)
q = lambda v: np.clip(np.nan_to_num(v * 255.999), 0, 255).astype(np.uint8).astype(int)
bad = np.abs(q(a) - q(b)).max(axis=2) > 1
print("bad", bad.sum(), "of", bad.size)
H, W = bad.shape
rays = np.array([rta.ray_for_pixel(cam, x, y) for y in range(H) for x in range(W)]).reshape(-1, 6)
pr = rta.probe(w, rays)
obj = np.array([p["object"] for p in pr]).reshape(H, W)
for y, x in zip(*np.nonzero(bad)):
    nb = obj[max(0,y-1):y+2, max(0,x-1):x+2]
    print((x, y), "obj", obj[y, x], "neighbour objs", sorted(set(nb.ravel().tolist())), "f64", np.round(a[y, x], 3), "f32", np.round(b[y, x], 3))
