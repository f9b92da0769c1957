"""Counts the steps of the bounding-sphere walk per camera wave.  Needs a library built with the
counters:  RTX_LIBNAME=librtx_count.so RTX_HIP_FLAGS=-DRTX_COUNT_WALK ./build.sh lib
then, on a GPU box:  RTX_LIB=$PWD/ray-tracer_amd/lib/librtx_count.so python3 tools/walk_count.py [hsize] [vsize]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
rta = load_package()
from ray_tracer_amd import scenes
W = int(sys.argv[1]) if len(sys.argv) > 1 else 960
H = int(sys.argv[2]) if len(sys.argv) > 2 else 540
w, cam = scenes.SCENES["random"](rta.api, hsize=W, vsize=H)
lib = ctypes.CDLL(os.environ["RTX_LIB"])
out = (ctypes.c_ulonglong * 8)()
prec = rta.F32 if os.environ.get("PREC") == "f32" else rta.F64  # each precision's kernels keep their own counters
read = lib.rtx_debug_walk_f32 if prec == rta.F32 else lib.rtx_debug_walk
for depth in (0, 1, 2, 5):
    read(out, 1)
    a = rta.render_rows(cam, w, depth=depth, precision=prec)
    assert read(out, 1) == 0
    waves = W * H / 64
    print("depth %d  per primary wave: group fetches %.1f  cluster fetches %.1f  object fetches %.1f  leaves %.1f   checksum %.6f"
          % (depth, out[0] / waves, out[1] / waves, out[2] / waves, out[3] / waves, float(a.sum())))
