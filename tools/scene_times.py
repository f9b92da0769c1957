#!/usr/bin/env python3
"""Device time of one 4K depth-5 frame of each benchmark scene (kernel_ms of rtx_render_device,
HIP events on the call's stream), for comparing builds (RTX_LIB) and settings quickly.

  python tools/scene_times.py [scene:precision ...]      e.g. glass:f64 random:f32
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402


def main():
    cases = sys.argv[1:] or ["default:f64", "default:f32", "spheres_on_plane:f64", "glass:f64", "glass:f32", "random:f64",
                             "random:f32"]
    rta = entry.load_package()
    from ray_tracer_amd import scenes
    W, H = int(os.environ.get("RTX_W", 3840)), int(os.environ.get("RTX_H", 2160))
    ctx = rta.Context(0)
    if os.environ.get("RTX_WORKSPACE_GIB"):
        ctx.set_workspace_limit(int(float(os.environ["RTX_WORKSPACE_GIB"]) * (1 << 30)))
    if os.environ.get("RTX_MEASURED_ABOVE"):  # worst-case queue bytes above which frames are sized level by level
        ctx.set_measured_queues(int(os.environ["RTX_MEASURED_ABOVE"]))
    for case in cases:
        name, precision = case.split(":")
        world, cam = scenes.SCENES[name](rta.api, hsize=W, vsize=H)
        flat = rta.flatten(world)
        prec = rta.F64 if precision == "f64" else rta.F32
        tile = torch.empty((H, W, 3), dtype=torch.float64 if prec == rta.F64 else torch.float32, device="cuda")
        nbytes = tile.numel() * tile.element_size()
        scene = rta.Scene(ctx, flat)
        cd = rta.camera_desc(cam)
        times, prim = [], []
        st = {}
        for _ in range(12):
            st = {}
            rta.render_device(ctx, scene, cd, 5, prec, 0, H, tile.data_ptr(), nbytes, None, st)
            times.append(st["kernel_ms"])
            prim.append(st["primary_kernel_ms"])
        rays = st["color_rays"] + st["shadow_rays"]
        ms = float(np.median(times[2:]))
        print("%-18s %s frame %8.4f ms (min %8.4f) primary %8.4f  chunks %d launches %d  %9.0f Mrays/s  queues %.2f GB  checksum %.12g" %
              (name, precision, ms, min(times), float(np.median(prim[2:])), st["n_chunks"], st["n_launches"], rays / ms / 1e3,
               ctx.queue_bytes() / 1e9, float(tile.double().sum())), flush=True)
        if os.environ.get("RTX_SHOW_LEVELS"):
            print("   rays per level", st["rays_per_level"], "shadow rays", st["shadow_rays"], flush=True)
        scene.close()


if __name__ == "__main__":
    main()
