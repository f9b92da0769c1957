#!/usr/bin/env python3
"""Benchmark of the MI355X render loop.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scene default] [--precision f64]

One step = one frame: every pixel's World::color_at at recursion depth 5, scene already
resident in HBM, framebuffer left in HBM (the first GPU's HBM when N > 1).  The headline
workload is the glass / mirror scene (reflection + refraction, Fresnel) at 3840x2160 in f64; the
other configurations are reported under "extra".  With N > 1 the frame is split into 10-row
bands, band b goes to GPU b mod N (every GPU gets its share of cheap and of dear rows), nothing
travels while the GPUs render and the bands are then assembled on the first GPU:
  * launched by torch.distributed.run (WORLD_SIZE set, one rank per GPU): one RCCL gather; the
    step time is the maximum over ranks; --gpus must equal WORLD_SIZE;
  * launched as a plain process with --gpus N: the N GPUs are driven from this one process
    through rtx_multi_* (one stream per GPU, strided peer copies into the first GPU).
Fewer than N visible GPUs is an error (RTX_BENCH_SAME_DEVICE=1 names GPU 0 N times instead: a
rehearsal of the code path on a one-GPU box, not a measurement).  Rank 0 prints one JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

# Algorithmic flops per ray-object test on the miss path and per shaded hit / primary ray:
# add, sub, mul, div, sqrt, pow count 1; neg, abs, compare, min, max, floor count 0; work the
# reference wastes (discarded 4th matrix row, `0.0 +`, repeated sqrt) is not counted.
F_TEST = {0: 54, 1: 12, 2: 45, 3: 49, 4: 55}  # sphere, plane, cube, cylinder, cone
F_PRIMARY = 36
F_SHADE = 130
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X FP64 vector peak (FMA counted as 2 flops)
HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E peak
FP32_VECTOR_PEAK_TFLOPS = 157.3  # packed FP32 vector peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scene", default="glass", choices=["default", "spheres_on_plane", "glass", "random", "random_glass"])
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--objects", type=int, default=10000, help="object count of the random scene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary scenes reported under 'extra'")
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="CPU work (seconds summed over threads) of the baseline sample")
    ap.add_argument("--ramp-ms", type=float, default=300.0,
                    help="untimed frames rendered before the warm-up steps until this much wall time has passed: the GPU "
                         "needs a few milliseconds of load to leave its idle clocks, more than W short frames give it")
    ap.add_argument("--gather", default="values", choices=["values", "rgba8"],
                    help="what the ranks send to rank 0: the frame's values, or 8-bit RGBA pixels converted on the GPU that "
                         "rendered them (src/scenes/canvas.rs:94-104), a sixth of the bytes in f64")
    return ap.parse_args()


def build_scene(rta, name, width, height, objects, n_lights=1):
    from ray_tracer_amd import scenes
    kw = dict(hsize=width, vsize=height)
    if name in ("random", "random_glass"):
        kw["n_objects"] = objects
    if n_lights != 1:
        kw["n_lights"] = n_lights
    return scenes.SCENES[name](rta.api, **kw)


def algorithmic_flops(shape_types, n_pixels, color_rays, shadow_rays):
    per_ray = float(sum(F_TEST[int(t)] for t in shape_types))
    return F_PRIMARY * n_pixels + per_ray * (color_rays + shadow_rays) + F_SHADE * shadow_rays


def host_threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit():
        n = min(n, int(env))
    return max(1, min(n, 64))


def cpu_baseline(scene_name, width, height, depth, objects, target_cpu_seconds):
    """The C restatement of the reference's render loop (oracle/), run reference-shaped:
    10-row bands (src/scenes/camera.rs:177) handed out dynamically to one thread per host core.
    The sample is about `target_cpu_seconds` of CPU work summed over the threads: whole frames
    repeated when a frame is cheaper than that, a centred block of bands when it is dearer."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    orc = entry.load_oracle()
    from ray_tracer_amd import scenes
    kw = dict(hsize=width, vsize=height)
    if scene_name in ("random", "random_glass"):
        kw["n_objects"] = objects
    world, cam = scenes.SCENES[scene_name](orc.api, **kw)
    threads = host_threads()
    full = np.zeros((height, width, 3))
    # calibrate on the band in the middle of the frame
    cnt = (C.c_uint64 * 2)(0, 0)
    mid = (height // 2 // 10) * 10
    t0 = time.perf_counter()
    orc._check(orc._render_rows(cam.h, world.h, depth, mid, min(mid + 10, height), full.ctypes.data_as(orc._dp), cnt))
    per_row = (time.perf_counter() - t0) / max(1, min(mid + 10, height) - mid)
    frame_cpu_s = per_row * height
    if frame_cpu_s >= target_cpu_seconds:
        rows = int(max(10 * threads, (target_cpu_seconds / per_row) // 10 * 10))
        rows = min(rows, height)
        r0 = ((height - rows) // 2 // 10) * 10
        repeats = 1
    else:
        r0, rows = 0, height
        repeats = int(min(1000, max(1, round(target_cpu_seconds / max(frame_cpu_s, 1e-9)))))
    bands = [(b, min(b + 10, r0 + rows)) for _ in range(repeats) for b in range(r0, r0 + rows, 10)]
    counters = [(C.c_uint64 * 2)(0, 0) for _ in bands]

    def work(i):
        a, b = bands[i]
        orc._check(orc._render_rows(cam.h, world.h, depth, a, b, full.ctypes.data_as(orc._dp), counters[i]))

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as pool:
        list(pool.map(work, range(len(bands))))
    dt = time.perf_counter() - t0
    rays = sum(c[0] + c[1] for c in counters)
    return {
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": "rows [%d, %d) of the %dx%d %s frame, depth %d, %d time(s): %d rays in %.2f s wall on %d threads; C "
                  "restatement of the reference's render loop in 10-row bands, not the Rust binary"
                  % (r0, r0 + rows, width, height, scene_name, depth, repeats, rays, dt, threads),
    }


def measured_traffic(scene_name, precision, width, height, kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of
    this workload (profiles/), or None when there is no such profile.  FETCH_SIZE and WRITE_SIZE
    count KiB; FETCH_SIZE is doubled as gfx950 tallies 128-byte read requests at 64 bytes."""
    for rnd in ("r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_%s_%dx%d_%s_pmc.json" % (rnd, scene_name, width, height, precision))
        if not os.path.exists(path):
            continue
        with open(path) as f:
            pmc = json.load(f)
        for kernel, counters in pmc.items():
            if kernel_prefix in kernel and "WRITE_SIZE" in counters and "FETCH_SIZE" in counters:
                return (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0, os.path.relpath(path, ROOT)
    return None, None


def launch_mode(args, environ):
    """How the N GPUs of --gpus are driven: ("single", 1), ("ranks", N) under torch.distributed.run
    (one rank per GPU; WORLD_SIZE must agree with --gpus), or ("multi", N) from this one process."""
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" in environ:
        world = int(environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit("bench.py: --gpus %d does not match WORLD_SIZE=%d of the launcher; the line would report the wrong "
                             "number of GPUs" % (args.gpus, world))
        return ("ranks", world) if world > 1 or environ.get("RTX_BENCH_DIST") == "1" else ("single", 1)
    if environ.get("RTX_BENCH_DIST") == "1" and args.gpus == 1:
        return ("ranks", 1)  # rehearsal of the multi-rank code path with one rank
    return ("multi", args.gpus) if args.gpus > 1 else ("single", 1)


def device_ids_for(n, visible, environ):
    """GPU ids of an N-GPU run in one process: 0..N-1, all of which must exist."""
    if visible >= n:
        return list(range(n))
    if environ.get("RTX_BENCH_SAME_DEVICE") == "1" and visible >= 1:
        return [0] * n  # rehearsal: the same GPU named n times
    raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible; refusing to report a %d-GPU number from fewer GPUs"
                     % (n, visible, n))


def bounce_flops(shape_types, n_pixels, st, primary_hits):
    """Algorithmic flops of the recursion levels below the primary rays of one frame."""
    per_ray = float(sum(F_TEST[int(t)] for t in shape_types))
    color = st["color_rays"] - n_pixels
    shadow = st["shadow_rays"] - primary_hits
    return per_ray * (color + shadow) + F_SHADE * shadow


def main():
    args = parse_args()
    mode, world_size = launch_mode(args, os.environ)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render loop has no CPU path")
    if mode == "multi":
        return main_multi(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    # RTX_BENCH_DIST=1 takes the multi-rank code path (process group, gather, reductions) with a
    # single rank too: a rehearsal of that path on a one-GPU box, not a measurement
    distributed = mode == "ranks"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # the single-rank rehearsal: any free port
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on stdout when its communicator comes up; stdout is for the one
        # JSON line, so the banner goes to stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    n_gpus = world_size
    rta = entry.load_package()
    from ray_tracer_amd import sharding

    prec = rta.F64 if args.precision == "f64" else rta.F32
    tdtype = torch.float64 if prec == rta.F64 else torch.float32
    world, cam = build_scene(rta, args.scene, args.width, args.height, args.objects)
    flat = rta.flatten(world)
    ctx = rta.Context(local_rank)
    # The timed steps pass a stream, and such calls size their ray queues for the worst case on the host
    # (they cannot wait for a level's lengths); the calls below that read counters and kernel times
    # return finished frames and would size them level by level instead.  One sizing for both, so that
    # the kernel times are those of the kernels the timed region ran.
    ctx.set_measured_queues((1 << 64) - 1)
    scene = rta.Scene(ctx, flat)
    cd = rta.camera_desc(cam)
    W, H = args.width, args.height
    # this rank's part: the bands b with b mod world_size == rank, back to back (the whole frame for one rank)
    rows = sharding.part_rows(H, world_size, rank)
    fmt = rta.PIXELS_RGBA8 if (distributed and args.gather == "rgba8") else 0
    part = torch.empty((rows, W, 4), dtype=torch.uint8, device="cuda") if fmt else torch.empty((rows, W, 3), dtype=tdtype, device="cuda")
    part_bytes = part.numel() * part.element_size()
    # One explicit HIP stream carries the render kernels, the timing events and (through
    # torch's stream bookkeeping) the ordering against the RCCL gather.
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    frame = torch.empty((H, W) + tuple(part.shape[2:]), dtype=part.dtype, device="cuda") if distributed and rank == 0 else None

    def render(stats=None, desc=cd, depth=args.depth):
        rta.render_part_device(ctx, scene, desc, depth, prec, sharding.BAND_SIZE, world_size, rank, fmt, part.data_ptr(),
                               part_bytes, stream, stats)

    def step():
        render()
        if distributed:
            return sharding.gather_parts(part, H, W, dst=0, out=frame)
        return part

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    # ray counts of this rank's part (deterministic per scene) and per-kernel device time; the primary
    # rays' own hits (a frame of depth 0 shades the same hits and casts nothing further)
    st, st0 = {}, {}
    render(st)
    render(st0, depth=0)
    primary_hits = st0["shadow_rays"]
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        for _ in range(5):
            render()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) / args.steps

    # the same loop without the gather: what the parts alone cost (reported beside the step time, not as it)
    render_only_wall = None
    if distributed:
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            render()
        fence()
        render_only_wall = time.perf_counter() - t1

    # the step once more with 8-bit RGBA pixels travelling instead of values (when that is not the step already)
    rgba8_step_wall = None
    if distributed and not fmt:
        part8 = torch.empty((rows, W, 4), dtype=torch.uint8, device="cuda")
        frame8 = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") if rank == 0 else None

        def step8():
            rta.render_part_device(ctx, scene, cd, args.depth, prec, sharding.BAND_SIZE, world_size, rank, rta.PIXELS_RGBA8,
                                   part8.data_ptr(), part8.numel(), stream)
            sharding.gather_parts(part8, H, W, dst=0, out=frame8)

        for _ in range(max(2, args.warmup)):
            step8()
        fence()
        t8 = time.perf_counter()
        for _ in range(args.steps):
            step8()
        fence()
        rgba8_step_wall = time.perf_counter() - t8

    # average device time of the kernels (HIP events around them on their own stream)
    k_primary, k_all, k_levels = [], [], []
    for _ in range(min(args.steps, 50)):
        s2 = {}
        render(s2)
        k_primary.append(s2["primary_kernel_ms"])
        k_all.append(s2["kernel_ms"])
        k_levels.append(s2["levels_ms"])

    single = not distributed
    moving_ms = host_ms = host_rgba8_ms = rgba8_ms = None
    if single:
        # A camera that moves between frames: a frame without secondary rays otherwise finds the
        # per-camera tables of the previous frame still valid and skips the launch that fills them.
        cams = [cd, moved_camera_desc(rta, cam)]
        for k in range(10):
            render(desc=cams[k & 1])
        fence()
        t_m = time.perf_counter()
        for k in range(args.steps):
            render(desc=cams[k & 1])
        fence()
        moving_ms = (time.perf_counter() - t_m) * 1e3 / args.steps
        # the frame as 8-bit RGBA pixels left in device memory (what the reference's GUI displays)
        rgba_dev = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
        for _ in range(5):
            rta.render_part_device(ctx, scene, cd, args.depth, prec, sharding.BAND_SIZE, 1, 0, rta.PIXELS_RGBA8, rgba_dev.data_ptr(),
                                   rgba_dev.numel(), stream)
        fence()
        t_r = time.perf_counter()
        for _ in range(args.steps):
            rta.render_part_device(ctx, scene, cd, args.depth, prec, sharding.BAND_SIZE, 1, 0, rta.PIXELS_RGBA8, rgba_dev.data_ptr(),
                                   rgba_dev.numel(), stream)
        fence()
        rgba8_ms = (time.perf_counter() - t_r) * 1e3 / args.steps
        # host-visible frame: render + copy to a host buffer (PCIe included), for the record
        # (buffers zero-filled first, as Canvas::new does: the copy is not charged their page faults)
        host = np.zeros((H, W, 3), dtype=np.float64 if prec == rta.F64 else np.float32)
        rgba = np.zeros((H, W, 4), dtype=np.uint8)
        host_ms = host_rgba8_ms = float("inf")
        for _ in range(3):
            t_h = time.perf_counter()
            rta._check(rta.rtx_render(ctx.h, scene.h, cd, args.depth, prec, 0, H, host.ctypes.data, host.nbytes, None, None, None))
            host_ms = min(host_ms, (time.perf_counter() - t_h) * 1e3)
            t_h = time.perf_counter()
            rta._check(rta.rtx_render_8bit(ctx.h, scene.h, cd, args.depth, prec, 0, H, 4, rgba.ctypes.data, rgba.nbytes, None, None, None))
            host_rgba8_ms = min(host_rgba8_ms, (time.perf_counter() - t_h) * 1e3)

    vals = torch.tensor([wall, float(st["color_rays"]), float(st["shadow_rays"]), render_only_wall or 0.0, rgba8_step_wall or 0.0],
                        dtype=torch.float64, device="cuda")
    if distributed:
        mx = vals.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = vals.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        wall = float(mx[0])
        render_only_wall = float(mx[3])
        rgba8_step_wall = float(mx[4]) if rgba8_step_wall is not None else None
        color_rays, shadow_rays = int(sm[1]), int(sm[2])
    else:
        color_rays, shadow_rays = st["color_rays"], st["shadow_rays"]

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        rays = color_rays + shadow_rays
        value = rays / (ms_per_step * 1e-3) / 1e6
        n_pix_part = rows * W
        elem = 8 if prec == rta.F64 else 4
        fb_bytes = n_pix_part * (4 if fmt else 3 * elem)
        flops_part = algorithmic_flops(flat["shape_type"], n_pix_part, st["color_rays"], st["shadow_rays"])
        single_kernel = st["n_launches"] == 1
        peak = FP64_VECTOR_PEAK_TFLOPS if prec == rta.F64 else FP32_VECTOR_PEAK_TFLOPS
        culled = bool(scene.features()["bounding_sphere_tests"])
        if single_kernel:
            # the frame is one launch of the camera kernel
            roof_kernel, launches = "k_primary_camera", 1
            k_ms = float(np.mean(k_primary))
            flops_launch = flops_part
        else:
            # the dominant kernel is the bounce kernel, one launch per recursion level: its own algorithmic flops
            # (everything below the primary rays) over its own time, per launch (the time between the levels'
            # events, which in scenes with regional levels includes the 5-9 us k_list_blocks launch of each level)
            feats = scene.features()
            roof_kernel = "k_bounce_tree" if feats["lane_walk"] else "k_bounce" if culled else "k_bounce_regional"
            launches = st["n_launches"] - 1
            k_ms = float(np.mean(k_levels)) / launches
            flops_launch = bounce_flops(flat["shape_type"], n_pix_part, st, primary_hits) / launches
        achieved = flops_launch / (k_ms * 1e-3) / 1e12
        traffic, traffic_source = (measured_traffic(args.scene, args.precision, W, H, roof_kernel) if n_gpus == 1 else (None, None))
        out = {
            "metric": "Mrays/s (primary+secondary) at 4K depth-5; frame ms at 1/2/4/8 GPUs",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {
                "workload": "%s scene %dx%d depth %d, %s" % (args.scene, W, H, args.depth, args.precision),
                "objects": int(len(flat["shape_type"])),
                "rays_per_frame": rays,
                "color_rays": color_rays,
                "shadow_rays": shadow_rays,
                "rays_per_level": st["rays_per_level"] if n_gpus == 1 else None,
                "parallelism": ("10-row bands dealt out to %d ranks + 1 gather of %s" % (n_gpus, "RGBA8 pixels" if fmt else "values"))
                if distributed else "single GPU",
            },
            "roofline": {
                "bound": "valu",
                "kernel": roof_kernel,
                "launches_per_frame": launches,
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                # HBM bytes per launch from the committed counter passes of this workload (not collected in this run)
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel_ms": k_ms,
                "algorithmic_flops_per_launch": flops_launch,
                "bounding_sphere_tests": culled,
                # with bounding-sphere tests most of the algorithm's ray/object tests are never executed:
                # "achieved" then counts work that was skipped and "frac" says nothing about the ALUs
                "frac_is_meaningful": not culled,
                # the whole frame (every launch) against the same roof
                "whole_frame": {"kernel_ms": float(np.mean(k_all)), "algorithmic_flops": flops_part,
                                "frac": flops_part / (float(np.mean(k_all)) * 1e-3) / 1e12 / peak},
                # the frame against the other roof: framebuffer bytes (the only traffic the algorithm needs) over the HBM peak
                "hbm": {"achieved": fb_bytes / (float(np.mean(k_all)) * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": fb_bytes / (float(np.mean(k_all)) * 1e-3) / 1e9 / HBM_PEAK_GBPS},
            },
            "frame": {
                "clock_ramp_ms_before_warmup": args.ramp_ms,
                "device_ms_per_frame": dev_ms,
                "camera_kernel_ms": float(np.mean(k_primary)),
                "levels_ms": float(np.mean(k_levels)),
                "color_only_Mrays_per_s": color_rays / (ms_per_step * 1e-3) / 1e6,
                "launches_per_frame": st["n_launches"],
                "queue_GB": ctx.queue_bytes() / 1e9,
                "queue_sizing": "worst case, on the host (calls on a caller's stream)",
            },
        }
        if single:
            out["frame"].update({
                # every frame seen from another camera position: the per-camera tables are rebuilt each time
                "moving_camera_ms": moving_ms,
                "moving_camera_Mrays_per_s": rays / (moving_ms * 1e-3) / 1e6,
                "rgba8_device_ms": rgba8_ms,
                "host_copy_inclusive_ms": host_ms,
                "host_rgba8_inclusive_ms": host_rgba8_ms,
            })
        if distributed:
            # slowest rank's part without the gather, and the bytes the gather moves to rank 0
            out["frame"]["render_only_ms"] = render_only_wall * 1e3 / args.steps
            out["frame"]["render_only_Mrays_per_s"] = rays / (render_only_wall / args.steps) / 1e6
            pixel_bytes = 4 if fmt else 3 * elem
            out["frame"]["gather_bytes"] = (H - sharding.part_rows(H, world_size, 0)) * W * pixel_bytes
            out["frame"]["gather_format"] = "rgba8" if fmt else args.precision
            if rgba8_step_wall is not None:
                # the same step when the ranks convert to RGBA8 before sending (a sixth of the f64 bytes)
                out["frame"]["rgba8_gather_ms_per_step"] = rgba8_step_wall * 1e3 / args.steps
                out["frame"]["rgba8_gather_Mrays_per_s"] = rays / (rgba8_step_wall / args.steps) / 1e6
                out["frame"]["rgba8_gather_bytes"] = (H - sharding.part_rows(H, world_size, 0)) * W * 4
        if not args.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline(args.scene, W, H, args.depth, args.objects, args.cpu_seconds)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if not args.no_extra and single and args.scene == "glass" and args.precision == "f64":
            out["extra"] = extra_scenes(rta, ctx, args)
        print(json.dumps(out))
    scene.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def main_multi(args):
    """--gpus N from one process: the frame on N GPUs through rtx_multi_* (what a single-process
    caller such as the reference's binary would use), assembled in the first GPU's memory."""
    n = args.gpus
    ids = device_ids_for(n, torch.cuda.device_count(), os.environ)
    rta = entry.load_package()
    prec = rta.F64 if args.precision == "f64" else rta.F32
    tdtype = torch.float64 if prec == rta.F64 else torch.float32
    world, cam = build_scene(rta, args.scene, args.width, args.height, args.objects)
    flat = rta.flatten(world)
    W, H = args.width, args.height
    fmt = rta.PIXELS_RGBA8 if args.gather == "rgba8" else 0
    torch.cuda.set_device(ids[0])
    frame = (torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") if fmt else torch.empty((H, W, 3), dtype=tdtype, device="cuda"))
    nbytes = frame.numel() * frame.element_size()
    multi = rta.Multi(ids)
    scene_h = multi.upload(flat)
    cd = rta.camera_desc(cam)

    def step(stats=None, flags=0):
        multi.render(scene_h, cd, args.depth, prec, fmt, frame.data_ptr(), nbytes, on_device=True, flags=flags, stats=stats)

    counted = {}
    step(counted, rta.MULTI_COUNT_RAYS)
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()  # returns when every GPU's bands are in the frame
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    render_ms, gather_ms = [], []
    for _ in range(min(args.steps, 20)):
        s2 = {}
        step(s2)
        render_ms.append(s2["render_ms"])
        gather_ms.append(s2["gather_ms"])
    ms_per_step = wall * 1e3 / args.steps
    rays = counted["color_rays"] + counted["shadow_rays"]
    flops = algorithmic_flops(flat["shape_type"], W * H, counted["color_rays"], counted["shadow_rays"])
    peak = (FP64_VECTOR_PEAK_TFLOPS if prec == rta.F64 else FP32_VECTOR_PEAK_TFLOPS) * n
    r_ms = float(np.mean(render_ms))
    out = {
        "metric": "Mrays/s (primary+secondary) at 4K depth-5; frame ms at 1/2/4/8 GPUs",
        "value": rays / (ms_per_step * 1e-3) / 1e6,
        "unit": "Mrays/s",
        "n_gpus": n,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {
            "workload": "%s scene %dx%d depth %d, %s" % (args.scene, W, H, args.depth, args.precision),
            "objects": int(len(flat["shape_type"])),
            "rays_per_frame": rays,
            "color_rays": counted["color_rays"],
            "shadow_rays": counted["shadow_rays"],
            "parallelism": "one process: 10-row bands dealt out to %d GPUs (rtx_multi), strided peer copies of %s into GPU %d"
                           % (n, "RGBA8 pixels" if fmt else "values", ids[0]),
            "device_ids": ids,
            "rehearsal_same_device": len(set(ids)) != len(ids),
        },
        "roofline": {
            "bound": "valu",
            "kernel": "k_primary_camera + k_bounce on every GPU (slowest GPU's part)",
            "achieved": flops / (r_ms * 1e-3) / 1e12,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": flops / (r_ms * 1e-3) / 1e12 / peak,
            "traffic": None,
            "kernel_ms": r_ms,
            "algorithmic_flops_per_launch": flops,
        },
        "frame": {
            "render_only_ms": r_ms,
            "gather_ms": float(np.mean(gather_ms)),
            "gather_bytes": int(s2["gather_bytes"]),
            "gather_format": "rgba8" if fmt else args.precision,
            "render_only_Mrays_per_s": rays / (r_ms * 1e-3) / 1e6,
        },
    }
    print(json.dumps(out))
    multi.close()


def moved_camera_desc(rta, cam):
    """The camera's rtx_camera_desc a quarter of a unit to the side (the fourth column of the
    inverse view transform is the camera's position): another origin, so other per-camera tables."""
    d = rta.camera_desc(cam)
    d.inverse_transform[3] += 0.25
    return d


def extra_scenes(rta, ctx, args):
    """The other single-GPU configurations, a few frames each (not the headline)."""
    res = {}
    # (the last one is the frame the multi-GPU configuration shards, here on one GPU: the reference shades with
    # lights[0] only, so its four lights cost what one does)
    cases = [("default", "f64", {}), ("default", "f32", {}), ("spheres_on_plane", "f64", {}), ("glass", "f64", {}), ("glass", "f32", {}),
             ("random", "f64", {"objects": 10000}), ("random", "f32", {"objects": 10000}),
             ("random_glass", "f64", {"objects": 10000}), ("random_glass", "f32", {"objects": 10000}),
             ("glass", "f64", {"width": 7680, "height": 4320, "n_lights": 4, "key": "glass_4lights_7680x4320_f64"})]
    never = (1 << 64) - 1

    def frame_time(name, precision, kw, measured_above=None):
        prec = rta.F64 if precision == "f64" else rta.F32
        tdtype = torch.float64 if prec == rta.F64 else torch.float32
        width, height = kw.get("width", args.width), kw.get("height", args.height)
        world, cam = build_scene(rta, name, width, height, kw.get("objects", args.objects), kw.get("n_lights", 1))
        flat = rta.flatten(world)
        own = rta.Context(torch.cuda.current_device())  # a context of its own: what it holds afterwards is this scene's
        if measured_above is not None:
            own.set_measured_queues(measured_above)
        scene = rta.Scene(own, flat)
        cd = rta.camera_desc(cam)
        tile = torch.empty((height, width, 3), dtype=tdtype, device="cuda")
        nbytes = tile.numel() * tile.element_size()
        st = {}
        rta.render_device(own, scene, cd, args.depth, prec, 0, height, tile.data_ptr(), nbytes, None, st)
        frames = 3 if st["kernel_ms"] > 50 else 20
        times, primary = [], []
        for _ in range(frames):
            s2 = {}
            rta.render_device(own, scene, cd, args.depth, prec, 0, height, tile.data_ptr(), nbytes, None, s2)
            times.append(s2["kernel_ms"])
            primary.append(s2["primary_kernel_ms"])
        culled = bool(scene.features()["bounding_sphere_tests"])
        queue_gb = own.queue_bytes() / 1e9
        scene.close()
        own.close()
        st["median_primary_kernel_ms"] = float(np.median(primary))
        return float(np.median(times)), st, flat, culled, queue_gb

    for name, precision, kw in cases:
        ms, st, flat, culled, queue_gb = frame_time(name, precision, kw)
        prec = rta.F64 if precision == "f64" else rta.F32
        rays = st["color_rays"] + st["shadow_rays"]
        flops = algorithmic_flops(flat["shape_type"], kw.get("height", args.height) * kw.get("width", args.width), st["color_rays"],
                                  st["shadow_rays"])
        key = kw.get("key", "%s_%s" % (name, precision))
        peak = FP64_VECTOR_PEAK_TFLOPS if prec == rta.F64 else FP32_VECTOR_PEAK_TFLOPS
        # A call that reports counters clears them and refills the per-camera tables first (k_frame_begin, which
        # the headline's timed loop skips while the camera stands still): frame_ms is from the start of that
        # launch to the end of the last.  camera_kernel_ms is the camera launch alone, and for a frame that is
        # this one launch the roofline fraction is taken over it, as the headline's is.
        k_ms = st["median_primary_kernel_ms"]
        roof_ms = k_ms if st["n_launches"] == 1 else ms
        res[key] = {
            "frame_ms": ms, "camera_kernel_ms": k_ms,
            "Mrays_per_s": rays / (ms * 1e-3) / 1e6, "rays": rays, "objects": int(len(flat["shape_type"])),
            "rays_per_level": st["rays_per_level"], "launches": st["n_launches"], "chunks": st["n_chunks"],
            # flops of the reference's algorithm (every ray against every object) per second: where the
            # bounding-sphere tests skip objects this exceeds what the ALUs execute, so no fraction of the
            # ALU peak is quoted there
            "valu_TFLOPs": flops / (roof_ms * 1e-3) / 1e12,
            "roofline_frac": None if culled else flops / (roof_ms * 1e-3) / 1e12 / peak,
            "bounding_sphere_tests": culled,
            # device memory the secondary-ray queues of this frame hold
            "queue_GB": queue_gb,
        }
        if name == "glass" and "key" not in kw:
            # the library's default for a frame this size is to size each level's queue from the lengths of
            # the level before (a host round trip per level); the same frame with every queue sized for the
            # worst case on the host, which is what a caller who passes a stream of its own gets
            ms_w, _, _, _, gb_w = frame_time(name, precision, kw, never)
            res["%s_%s" % (name, precision)]["worst_case_queues"] = {"frame_ms": ms_w, "queue_GB": gb_w,
                                                                      "Mrays_per_s": rays / (ms_w * 1e-3) / 1e6}
    if not args.no_cpu_baseline:
        # the oracle beside the other two scenes as well (short samples: a few seconds of CPU work each)
        for name in ("default", "random"):
            base = cpu_baseline(name, args.width, args.height, args.depth, 10000, min(args.cpu_seconds, 6.0))
            res["%s_f64" % name]["cpu_baseline"] = base
            res["%s_f64" % name]["gpu_over_cpu"] = res["%s_f64" % name]["Mrays_per_s"] / base["value"]
    return res


if __name__ == "__main__":
    main()
