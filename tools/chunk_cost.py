#!/usr/bin/env python3
"""What splitting a frame into row chunks costs: the glass scene at 4K rendered under several
workspace limits (the queues of one chunk must fit the limit).  Prints chunks, launches and the
device time of the frame per limit.

  python tools/chunk_cost.py [scene] [precision] [limit_GiB ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402


def main():
    scene_name = sys.argv[1] if len(sys.argv) > 1 else "glass"
    precision = sys.argv[2] if len(sys.argv) > 2 else "f64"
    limits = [float(v) for v in sys.argv[3:]] or [48.0, 16.0, 8.0, 4.0, 2.0, 1.0, 0.5]
    rta = entry.load_package()
    from ray_tracer_amd import scenes
    W, H = 3840, 2160
    world, cam = scenes.SCENES[scene_name](rta.api, hsize=W, vsize=H)
    flat = rta.flatten(world)
    prec = rta.F64 if precision == "f64" else rta.F32
    tile = torch.empty((H, W, 3), dtype=torch.float64 if prec == rta.F64 else torch.float32, device="cuda")
    nbytes = tile.numel() * tile.element_size()
    for gib in limits:
        ctx = rta.Context(0)
        ctx.set_workspace_limit(int(gib * (1 << 30)))
        scene = rta.Scene(ctx, flat)
        cd = rta.camera_desc(cam)
        times = []
        st = {}
        for _ in range(8):
            st = {}
            rta.render_device(ctx, scene, cd, 5, prec, 0, H, tile.data_ptr(), nbytes, None, st)
            times.append(st["kernel_ms"])
        print("limit %5.1f GiB: chunks %3d launches %3d frame %.3f ms (min %.3f)" %
              (gib, st["n_chunks"], st["n_launches"], float(np.median(times[2:])), min(times)), flush=True)
        scene.close()
        ctx.close()


if __name__ == "__main__":
    main()
