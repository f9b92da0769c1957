"""Cross-check of the algorithmic-flop constants bench.py prices a frame with: renders small
frames with the operation-counting build of the oracle (oracle/opcount.cpp) and compares the
adds, multiplies, divisions, square roots and pows it performed with bench.algorithmic_flops().
The oracle does the reference's dead work too (fourth matrix rows, the `0.0 +` that starts each
dot product, intersections computed twice), so it must count MORE than the formula, which
prices the miss path of each test only.  CPU only:  python3 tools/flop_check.py"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libopcount.so")
subprocess.run(["g++", "-O1", "-fpermissive", "-w", "-ffp-contract=off", "-fPIC", "-shared", "-pthread",
                os.path.join(ROOT, "oracle", "opcount.cpp"), "-o", LIB], check=True)
os.environ["ORC_LIB"] = LIB
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
rta = load_package()  # host mirror only (scene flattening); nothing here touches a GPU
from ray_tracer_amd import scenes  # noqa: E402

ops = (C.c_uint64 * 6)()
orc.lib.opc_read.argtypes = [C.POINTER(C.c_uint64), C.c_int]
orc.lib.opc_read.restype = None

print("%-18s %9s %12s %14s %14s %7s   %11s %11s %9s %8s %7s" % ("scene", "pixels", "rays", "counted", "formula", "ratio",
                                                                "add/sub", "mul", "div", "sqrt", "pow"))
for name, kw in [("default", dict(hsize=160, vsize=90)), ("spheres_on_plane", dict(hsize=160, vsize=90)),
                 ("glass", dict(hsize=96, vsize=54)), ("random", dict(hsize=48, vsize=27, n_objects=300))]:
    w, cam = scenes.SCENES[name](orc.api, **kw)
    shape_types = list(rta.flatten(scenes.SCENES[name](rta.api, **kw)[0])["shape_type"])
    orc.lib.opc_read(ops, 1)
    cnt = [0, 0]
    orc.render_rows(cam, w, counters=cnt)
    orc.lib.opc_read(ops, 1)
    counted = sum(ops[k] for k in range(5))
    formula = bench.algorithmic_flops(shape_types, kw["hsize"] * kw["vsize"], cnt[0], cnt[1])
    print("%-18s %9d %12d %14d %14d %7.2f   %11d %11d %9d %8d %7d" % (name, kw["hsize"] * kw["vsize"], cnt[0] + cnt[1], counted,
                                                                  int(formula), counted / formula, ops[0], ops[1], ops[2], ops[3], ops[4]))
