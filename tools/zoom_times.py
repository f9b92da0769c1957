#!/usr/bin/env python3
"""The glass scene seen through a narrow field of view (a window of the 4K frame at the 4K frame's
pixel density), to measure what the rays of one screen region cost when they are traced together.

  python tools/zoom_times.py W H zoom to_x to_y to_z [precision]
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402


def main():
    W, H, zoom = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    to = tuple(float(v) for v in sys.argv[4:7])
    precision = sys.argv[7] if len(sys.argv) > 7 else "f64"
    rta = entry.load_package()
    from ray_tracer_amd import scenes
    ctx = rta.Context(0)
    world, _ = scenes.glass_scene(rta.api, hsize=W, vsize=H)
    # the 4K camera's pixel is tan(fov/2) * 2 / 3840 wide at distance 1; keep that with W pixels across
    half = math.tan(math.pi / 6.0) * W / 3840.0 / zoom
    cam = rta.api.camera(W, H, 2.0 * math.atan(half))
    cam.set_view((0.0, 2.2, -6.5), to, (0.0, 1.0, 0.0))
    flat = rta.flatten(world)
    prec = rta.F64 if precision == "f64" else rta.F32
    tile = torch.empty((H, W, 3), dtype=torch.float64 if prec == rta.F64 else torch.float32, device="cuda")
    nbytes = tile.numel() * tile.element_size()
    scene = rta.Scene(ctx, flat)
    cd = rta.camera_desc(cam)
    times = []
    st = {}
    for _ in range(12):
        st = {}
        rta.render_device(ctx, scene, cd, 5, prec, 0, H, tile.data_ptr(), nbytes, None, st)
        times.append(st["kernel_ms"])
    rays = st["color_rays"] + st["shadow_rays"]
    ms = float(np.median(times[2:]))
    print("zoom %g %dx%d %s frame %8.4f ms levels %8.4f  %9.0f Mrays/s  rays per level %s" %
          (zoom, W, H, precision, ms, st["levels_ms"], rays / ms / 1e3, st["rays_per_level"]), flush=True)


if __name__ == "__main__":
    main()
