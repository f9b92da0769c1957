"""Regenerates tests/golden/frames.npz from the CPU oracle: python3 tests/golden/make_golden.py

The reference (Rust) cannot be built in this environment, so these frames are NOT outputs of
the reference binary: they are outputs of oracle/rt_oracle.c, whose arithmetic is pinned to the
reference's own unit-test constants by tests/test_oracle_golden.py.  They freeze the oracle, so
that a change in it shows up as a diff here, and give the GPU tests a fixture that does not
depend on the oracle being rebuilt identically on another machine."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import orc  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

entry.load_package()
from ray_tracer_amd import scenes  # noqa: E402

# name -> (scene, kwargs, depth)
CASES = {
    "default_50x40_d5": ("default", dict(hsize=50, vsize=40), 5),
    "spheres_on_plane_48x27_d5": ("spheres_on_plane", dict(hsize=48, vsize=27), 5),
    "glass_64x36_d5": ("glass", dict(hsize=64, vsize=36), 5),
    "glass_64x36_d2": ("glass", dict(hsize=64, vsize=36), 2),
    "random60_48x27_d5": ("random", dict(n_objects=60, hsize=48, vsize=27), 5),
    "grouped_80x45_d5": ("grouped", dict(hsize=80, vsize=45), 5),
}


def render(name):
    scene, kw, depth = CASES[name]
    w, cam = scenes.SCENES[scene](orc.api, **kw)
    cnt = [0, 0]  # color_at calls, is_shadowed calls
    img = orc.render_rows(cam, w, depth=depth, counters=cnt)
    return img, np.array(cnt, dtype=np.uint64)


if __name__ == "__main__":
    out = {}
    for name in CASES:
        img, cnt = render(name)
        out[name] = img
        out[name + "_rays"] = cnt
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "frames.npz"), **out)
    print("wrote", sorted(out))
