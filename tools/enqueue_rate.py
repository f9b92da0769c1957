"""Host-side cost of enqueueing one frame versus its device time (default scene, 4K, f64)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
rta = load_package()
from ray_tracer_amd import scenes
w, cam = scenes.default_scene(rta.api, 3840, 2160)
ctx = rta.Context(0)
scene = rta.Scene(ctx, rta.flatten(w))
cd = rta.camera_desc(cam)
tile = torch.empty((2160, 3840, 3), dtype=torch.float64, device="cuda")
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
n = tile.numel() * 8
for reps in (200, 1000):
    for _ in range(20):
        rta.render_device(ctx, scene, cd, 5, rta.F64, 0, 2160, tile.data_ptr(), n, s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        rta.render_device(ctx, scene, cd, 5, rta.F64, 0, 2160, tile.data_ptr(), n, s.cuda_stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("reps %d: enqueue %.1f us/frame, total %.1f us/frame" % (reps, (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6))
