"""f32 fast path on the randomised worlds of tests/test_fuzz_parity.py: share of pixels within 1/255."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
rta = load_package()
import test_fuzz_parity as tf
q = lambda v: np.clip(np.nan_to_num(v * 255.999), 0, 255).astype(np.uint8).astype(int)
worst = []
scale = int(os.environ.get("SCALE", "4"))
only = os.environ.get("SEED")
for seed in ([int(only)] if only else range(int(sys.argv[1]) if len(sys.argv) > 1 else 24)):
    w, cam = tf.build_world(rta.api, 1000 + seed, seed % 2 == 1, res_scale=scale)
    a = rta.render_rows(cam, w, precision=rta.F64)
    b = rta.render_rows(cam, w, precision=rta.F32)
    bad = np.abs(q(a) - q(b)).max(axis=2) > 1
    ok = 1.0 - bad.mean()
    worst.append((ok, seed, a.shape))
    print("seed %2d %s: %.4f" % (seed, a.shape[:2], ok))
    if only:
        H, W = bad.shape
        rays = np.array([rta.ray_for_pixel(cam, x, y) for y in range(0, H, 2) for x in range(0, W, 2)])
        obj = rta.probe(w, rays)["object"].reshape((H + 1) // 2, (W + 1) // 2)
        sub = bad[::2, ::2]
        for o in np.unique(obj):
            m = obj == o
            print("  primary object %3d: %6d px, %5d bad" % (o, m.sum(), (sub & m).sum()))
        full = rta.probe(w, np.array([rta.ray_for_pixel(cam, x, y) for y in range(H) for x in range(W)]))
        fo = full["object"].reshape(H, W)
        ys, xs = np.nonzero(bad)
        for y, x in list(zip(ys, xs))[:: max(1, len(ys) // 25)]:
            print("  (%3d,%3d) obj %2d t %.4f inside %d shadowed %d  f64 %s  f32 %s" % (x, y, fo[y, x], full["t"].reshape(H, W)[y, x],
                  full["inside"].reshape(H, W)[y, x], full["shadowed"].reshape(H, W)[y, x], np.round(a[y, x], 3), np.round(b[y, x], 3)))
        for depth in (0, 1, 2):
            a_d = rta.render_rows(cam, w, depth=depth, precision=rta.F64)
            b_d = rta.render_rows(cam, w, depth=depth, precision=rta.F32)
            print("  depth %d: %d bad" % (depth, (np.abs(q(a_d) - q(b_d)).max(axis=2) > 1).sum()))
        for j in range(0, 0, max(1, H // 40)):
            print("".join(" .:*#"[min(4, int(bad[j:j + max(1, H // 40), i:i + max(1, W // 100)].mean() * 8 + (bad[j:j + max(1, H // 40), i:i + max(1, W // 100)].any())))] for i in range(0, W, max(1, W // 100))))
print("min", min(worst))
