"""Committed golden frames (tests/golden/frames.npz, written by tests/golden/make_golden.py
from the CPU oracle) against the oracle as built here, and against the HIP path."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)
FRAMES = np.load(os.path.join(HERE, "golden", "frames.npz"))


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_oracle_reproduces_golden_frames(name):
    """Same compiler flags, same arithmetic: the only latitude is libm's pow (<= 1 ulp), which
    reaches a colour through the specular term only."""
    img, rays = make_golden.render(name)
    assert img.shape == FRAMES[name].shape
    assert np.abs(img - FRAMES[name]).max() <= 1e-12
    assert np.array_equal(rays, FRAMES[name + "_rays"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_hip_path_matches_golden_frames(rta, name):
    """f64 colours within 1e-9 per channel; ray counters equal."""
    from ray_tracer_amd import scenes
    scene, kw, depth = make_golden.CASES[name]
    w, cam = scenes.SCENES[scene](rta.api, **kw)
    st = {}
    img = rta.render_rows(cam, w, depth=depth, stats=st)
    assert np.abs(img - FRAMES[name]).max() <= 1e-9
    assert [st["color_rays"], st["shadow_rays"]] == [int(v) for v in FRAMES[name + "_rays"]]
