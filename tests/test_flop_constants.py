"""The algorithmic-flop formula bench.py prices a frame with must not overstate the work: the
oracle built with an operation-counting scalar type (oracle/opcount.cpp) performs the
reference's arithmetic literally, dead work included, so its count of adds, multiplies,
divisions, square roots and pows is an upper bound of the formula -- and not a distant one."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_formula_is_a_lower_bound_of_the_counted_operations():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "flop_check.py")], check=True, capture_output=True,
                         text=True, cwd=ROOT).stdout
    rows = [l.split() for l in out.splitlines()[1:] if l.strip()]
    assert [r[0] for r in rows] == ["default", "spheres_on_plane", "glass", "random"]
    for r in rows:
        counted, formula = int(r[3]), int(r[4])
        assert formula <= counted <= 3 * formula, r
