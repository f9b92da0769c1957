"""The screen-region arithmetic of camera launches (ray-tracer_amd/csrc/rt_launch.h: region_split, region_magic
and the interleaving of rows over rows of regions), checked on the host for a range of frame shapes: it decides
which sub-queue the children of a camera ray go to, and the queue planner sizes those sub-queues by the same
numbers (the band loop it serves: src/scenes/camera.rs:177-193)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_region_arithmetic_on_the_host():
    src = os.path.join(ROOT, "tests", "native", "region_map_check.cpp")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "region_map_check")
        subprocess.run(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok ")
