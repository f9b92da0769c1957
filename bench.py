#!/usr/bin/env python3
"""Benchmark of the MI355X render loop.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scene default] [--precision f64]

One step = one frame: every pixel's World::color_at at recursion depth 5, scene already
resident in HBM, framebuffer left in HBM (rank 0's HBM after the gather when N > 1).  The
headline workload is the reference's default world at 3840x2160 in f64.  With N > 1 (launched
by torch.distributed.run, one rank per GPU) the frame is split into row tiles, each rank
renders its tile and one RCCL gather assembles the frame on rank 0; the step time is the
maximum over ranks.  Rank 0 prints one JSON line.
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
    # a frame of the headline scene takes ~70 us: 2000 of them amortise the ~2 ms the GPU needs to
    # come out of idle after the fence that precedes the timed region
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--scene", default="default", choices=["default", "spheres_on_plane", "glass", "random"])
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--objects", type=int, default=10000, help="object count of the random scene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary scenes reported under 'extra'")
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="CPU work (seconds summed over threads) of the baseline sample")
    return ap.parse_args()


def build_scene(rta, name, width, height, objects):
    from ray_tracer_amd import scenes
    kw = dict(hsize=width, vsize=height)
    if name == "random":
        kw["n_objects"] = objects
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
    if scene_name == "random":
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
    path = os.path.join(ROOT, "profiles", "r01_%s_%dx%d_%s_pmc.json" % (scene_name, width, height, precision))
    if not os.path.exists(path):
        return None
    with open(path) as f:
        pmc = json.load(f)
    for kernel, counters in pmc.items():
        if kernel_prefix in kernel and "WRITE_SIZE" in counters and "FETCH_SIZE" in counters:
            return (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0
    return None


def main():
    args = parse_args()
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render loop has no CPU path")
    torch.cuda.set_device(local_rank)
    # RTX_BENCH_DIST=1 takes the multi-rank code path (process group, gather, reductions) with a
    # single rank too: a rehearsal of that path on a one-GPU box, not a measurement
    distributed = world_size > 1 or os.environ.get("RTX_BENCH_DIST") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world_size
    rta = entry.load_package()
    from ray_tracer_amd import sharding

    prec = rta.F64 if args.precision == "f64" else rta.F32
    tdtype = torch.float64 if prec == rta.F64 else torch.float32
    world, cam = build_scene(rta, args.scene, args.width, args.height, args.objects)
    flat = rta.flatten(world)
    ctx = rta.Context(local_rank)
    scene = rta.Scene(ctx, flat)
    cd = rta.camera_desc(cam)
    tiles = sharding.row_tiles(args.height, world_size)
    r0, r1 = tiles[rank]
    tile = torch.empty((r1 - r0, args.width, 3), dtype=tdtype, device="cuda")
    # One explicit HIP stream carries the render kernels, the timing events and (through
    # torch's stream bookkeeping) the ordering against the RCCL gather.  A null handle would
    # mean "the context's private stream" to rtx_render_device, which torch knows nothing about.
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    frame = torch.empty((args.height, args.width, 3), dtype=tdtype, device="cuda") if distributed and rank == 0 else None

    def step():
        rta.render_device(ctx, scene, cd, args.depth, prec, r0, r1, tile.data_ptr(), tile.numel() * tile.element_size(), stream)
        if distributed:
            return sharding.gather_frame(tile, tiles, args.width, dst=0, out=frame)
        return tile

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    # ray counts of this rank's tile (deterministic per scene) and per-kernel device time
    st = {}
    rta.render_device(ctx, scene, cd, args.depth, prec, r0, r1, tile.data_ptr(), tile.numel() * tile.element_size(), stream, st)
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

    # the same loop without the gather: what the tiles alone cost (reported beside the step time, not as it)
    render_only_wall = None
    if distributed:
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            rta.render_device(ctx, scene, cd, args.depth, prec, r0, r1, tile.data_ptr(), tile.numel() * tile.element_size(), stream)
        fence()
        render_only_wall = time.perf_counter() - t1

    # average device time of the dominant kernel (HIP events around it on its own stream)
    k_primary, k_all = [], []
    for _ in range(min(args.steps, 50)):
        s2 = {}
        rta.render_device(ctx, scene, cd, args.depth, prec, r0, r1, tile.data_ptr(), tile.numel() * tile.element_size(), stream, s2)
        k_primary.append(s2["primary_kernel_ms"])
        k_all.append(s2["kernel_ms"])
    # host-visible frame: render + copy to a host buffer (PCIe included), for the record
    # (buffers zero-filled first, as Canvas::new does: the copy is not charged their page faults)
    host = np.zeros((r1 - r0, args.width, 3), dtype=np.float64 if prec == rta.F64 else np.float32)
    rgba = np.zeros((r1 - r0, args.width, 4), dtype=np.uint8)
    host_ms = host_rgba8_ms = float("inf")
    for _ in range(3):
        t_h = time.perf_counter()
        rta._check(rta.rtx_render(ctx.h, scene.h, cd, args.depth, prec, r0, r1, host.ctypes.data, host.nbytes, None, None, None))
        host_ms = min(host_ms, (time.perf_counter() - t_h) * 1e3)
        t_h = time.perf_counter()
        rta._check(rta.rtx_render_8bit(ctx.h, scene.h, cd, args.depth, prec, r0, r1, 4, rgba.ctypes.data, rgba.nbytes, None, None, None))
        host_rgba8_ms = min(host_rgba8_ms, (time.perf_counter() - t_h) * 1e3)

    vals = torch.tensor([wall, float(st["color_rays"]), float(st["shadow_rays"]), render_only_wall or 0.0], dtype=torch.float64,
                        device="cuda")
    if distributed:
        mx = vals.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = vals.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        wall = float(mx[0])
        render_only_wall = float(mx[3])
        color_rays, shadow_rays = int(sm[1]), int(sm[2])
    else:
        color_rays, shadow_rays = st["color_rays"], st["shadow_rays"]

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        rays = color_rays + shadow_rays
        value = rays / (ms_per_step * 1e-3) / 1e6
        n_pix_tile = (r1 - r0) * args.width
        flops_tile = algorithmic_flops(flat["shape_type"], n_pix_tile, st["color_rays"], st["shadow_rays"])
        single_kernel = st["n_launches"] == 1
        k_ms = float(np.mean(k_primary)) if single_kernel else float(np.mean(k_all))
        peak = FP64_VECTOR_PEAK_TFLOPS if prec == rta.F64 else FP32_VECTOR_PEAK_TFLOPS
        achieved = flops_tile / (k_ms * 1e-3) / 1e12
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
                "workload": "%s scene %dx%d depth %d, %s" % (args.scene, args.width, args.height, args.depth, args.precision),
                "objects": int(len(flat["shape_type"])),
                "rays_per_frame": rays,
                "color_rays": color_rays,
                "shadow_rays": shadow_rays,
                "parallelism": "row tiles x%d + 1 gather" % n_gpus if n_gpus > 1 else "single GPU",
            },
            "roofline": {
                "bound": "valu",
                "kernel": "k_primary_camera" if single_kernel else "k_primary_camera + k_bounce (all levels)",
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                "traffic": measured_traffic(args.scene, args.precision, args.width, args.height, "k_primary_camera")
                if single_kernel and n_gpus == 1 else None,
                "kernel_ms": k_ms,
                "algorithmic_flops_per_launch": flops_tile,
                "bounding_sphere_tests": scene.features()["bounding_sphere_tests"],
                # with bounding-sphere tests most of the algorithm's ray/object tests are never executed:
                # "achieved" then counts work that was skipped and "frac" says nothing about the ALUs
                "frac_is_meaningful": not scene.features()["bounding_sphere_tests"],
                "hbm_write_GBps": n_pix_tile * 3 * tile.element_size() / (k_ms * 1e-3) / 1e9,
                # the same launch against the other roof: framebuffer bytes (the only HBM traffic) over the HBM peak
                "hbm": {"achieved": n_pix_tile * 3 * tile.element_size() / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": n_pix_tile * 3 * tile.element_size() / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
            },
            "frame": {
                "device_ms_per_frame": dev_ms,
                "color_only_Mrays_per_s": color_rays / (ms_per_step * 1e-3) / 1e6,
                "host_copy_inclusive_ms": host_ms,
                "host_rgba8_inclusive_ms": host_rgba8_ms,
                "launches_per_frame": st["n_launches"],
            },
        }
        if distributed:
            # slowest rank's tile without the gather, and the frame bytes the gather moves to rank 0
            out["frame"]["render_only_ms"] = render_only_wall * 1e3 / args.steps
            out["frame"]["render_only_Mrays_per_s"] = rays / (render_only_wall / args.steps) / 1e6
            out["frame"]["gather_bytes"] = args.height * args.width * 3 * tile.element_size() * (n_gpus - 1) // max(n_gpus, 1)
        if not args.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline(args.scene, args.width, args.height, args.depth, args.objects, args.cpu_seconds)
        if not args.no_extra and n_gpus == 1 and args.scene == "default" and args.precision == "f64":
            out["extra"] = extra_scenes(rta, ctx, args)
        print(json.dumps(out))
    scene.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def extra_scenes(rta, ctx, args):
    """The other single-GPU configurations, a few frames each (not the headline)."""
    res = {}
    cases = [("default", "f32", {}), ("spheres_on_plane", "f64", {}), ("glass", "f64", {}), ("glass", "f32", {}),
             ("random", "f64", {"objects": 10000}), ("random", "f32", {"objects": 10000})]
    for name, precision, kw in cases:
        prec = rta.F64 if precision == "f64" else rta.F32
        tdtype = torch.float64 if prec == rta.F64 else torch.float32
        world, cam = build_scene(rta, name, args.width, args.height, kw.get("objects", args.objects))
        flat = rta.flatten(world)
        scene = rta.Scene(ctx, flat)
        cd = rta.camera_desc(cam)
        tile = torch.empty((args.height, args.width, 3), dtype=tdtype, device="cuda")
        nbytes = tile.numel() * tile.element_size()
        st = {}
        rta.render_device(ctx, scene, cd, args.depth, prec, 0, args.height, tile.data_ptr(), nbytes, None, st)
        frames = 3 if st["kernel_ms"] > 50 else 20
        times = []
        for _ in range(frames):
            s2 = {}
            rta.render_device(ctx, scene, cd, args.depth, prec, 0, args.height, tile.data_ptr(), nbytes, None, s2)
            times.append(s2["kernel_ms"])
        ms = float(np.median(times))
        rays = st["color_rays"] + st["shadow_rays"]
        flops = algorithmic_flops(flat["shape_type"], args.height * args.width, st["color_rays"], st["shadow_rays"])
        peak = FP64_VECTOR_PEAK_TFLOPS if prec == rta.F64 else FP32_VECTOR_PEAK_TFLOPS
        culled = bool(scene.features()["bounding_sphere_tests"])
        res["%s_%s" % (name, precision)] = {
            "frame_ms": ms, "Mrays_per_s": rays / (ms * 1e-3) / 1e6, "rays": rays, "objects": int(len(flat["shape_type"])),
            "rays_per_level": st["rays_per_level"], "launches": st["n_launches"], "chunks": st["n_chunks"],
            # flops of the reference's algorithm (every ray against every object) per second: where the
            # bounding-sphere tests skip objects this exceeds what the ALUs execute, so no fraction of the
            # ALU peak is quoted there
            "valu_TFLOPs": flops / (ms * 1e-3) / 1e12,
            "roofline_frac": None if culled else flops / (ms * 1e-3) / 1e12 / peak,
            "bounding_sphere_tests": culled,
        }
        scene.close()
    return res


if __name__ == "__main__":
    main()
