"""bench.py --gpus N: the number of GPUs it reports is the number it used.

The band loop of Camera::parallel_render (src/scenes/camera.rs:177-193) is what the N GPUs share;
a line that says n_gpus: 1 for a --gpus 2 run, or a 2-GPU line measured on one GPU, would be a
wrong measurement, so both are refused."""
import argparse
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(gpus):
    return argparse.Namespace(gpus=gpus)


def test_launch_mode_follows_gpus_and_the_launcher():
    import bench
    assert bench.launch_mode(_args(1), {}) == ("single", 1)
    assert bench.launch_mode(_args(2), {}) == ("multi", 2)          # plain process: N GPUs from this process
    assert bench.launch_mode(_args(8), {}) == ("multi", 8)
    assert bench.launch_mode(_args(2), {"WORLD_SIZE": "2"}) == ("ranks", 2)  # torch.distributed.run: one rank per GPU
    assert bench.launch_mode(_args(1), {"WORLD_SIZE": "1"}) == ("single", 1)
    assert bench.launch_mode(_args(1), {"RTX_BENCH_DIST": "1"}) == ("ranks", 1)
    for gpus, world in ((2, "1"), (1, "2"), (4, "8")):
        with pytest.raises(SystemExit, match="does not match WORLD_SIZE"):
            bench.launch_mode(_args(gpus), {"WORLD_SIZE": world})
    with pytest.raises(SystemExit):
        bench.launch_mode(_args(0), {})


def test_too_few_visible_gpus_is_an_error_not_a_smaller_run():
    import bench
    assert bench.device_ids_for(2, 8, {}) == [0, 1]
    assert bench.device_ids_for(8, 8, {}) == list(range(8))
    with pytest.raises(SystemExit, match="only 1 GPU"):
        bench.device_ids_for(2, 1, {})
    with pytest.raises(SystemExit):
        bench.device_ids_for(2, 0, {"RTX_BENCH_SAME_DEVICE": "1"})
    assert bench.device_ids_for(3, 1, {"RTX_BENCH_SAME_DEVICE": "1"}) == [0, 0, 0]  # rehearsal knob


def _run_bench(extra_args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RTX_BENCH_DIST", "RTX_BENCH_SAME_DEVICE")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--ramp-ms", "0", "--width", "320",
                           "--height", "180", "--no-cpu-baseline", "--no-extra"] + extra_args,
                          env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.gpu
def test_gpus_2_on_fewer_gpus_fails_and_the_rehearsal_reports_2():
    import torch
    if torch.cuda.device_count() < 2:
        r = _run_bench(["--gpus", "2"], {})
        assert r.returncode != 0 and "only 1 GPU" in r.stderr
    r = _run_bench(["--gpus", "2"], {"RTX_BENCH_SAME_DEVICE": "1"} if torch.cuda.device_count() < 2 else {})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2
    assert line["frame"]["render_only_ms"] > 0 and line["frame"]["gather_bytes"] > 0
    assert line["config"]["rehearsal_same_device"] == (torch.cuda.device_count() < 2)
    # and the single-GPU line names the glass scene as its workload and carries the roofline of the bounce kernel
    r = _run_bench([], {})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["workload"].startswith("glass scene")
    assert line["roofline"]["kernel"] == "k_bounce_regional" and line["roofline"]["launches_per_frame"] == 5
    assert 0 < line["roofline"]["frac"] < 1
