"""Largest colour difference between the HIP f64 path and the committed oracle frames:
python3 tools/golden_diff.py   (on a GPU box)"""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rta = load_package()
from ray_tracer_amd import scenes
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
frames = np.load(os.path.join(ROOT, "tests", "golden", "frames.npz"))
for name in sorted(mg.CASES):
    scene, kw, depth = mg.CASES[name]
    w, cam = scenes.SCENES[scene](rta.api, **kw)
    img = rta.render_rows(cam, w, depth=depth)
    d = np.abs(img - frames[name])
    print("%-28s max |diff| %.3e   pixels differing at all %6.2f %%" % (name, d.max(), 100.0 * (d.max(axis=2) > 0).mean()))
