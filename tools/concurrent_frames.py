"""Do frames rendered side by side on separate streams overlap?  Renders the same frame from 1, 2
and 4 host threads (a context and a device buffer each) and prints frames per second.
python3 tools/concurrent_frames.py [scene] [f64|f32] [frames per thread]   (on a GPU box)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
rta = load_package()
from ray_tracer_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "random"
prec = rta.F32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else rta.F64
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
W, H = 3840, 2160
w, cam = scenes.SCENES[name](rta.api, hsize=W, vsize=H)
flat = rta.flatten(w)
cd = rta.camera_desc(cam)
dtype = torch.float64 if prec == rta.F64 else torch.float32


def worker(ctx, scene, buf, n):
    for _ in range(n):
        rta.render_device(ctx, scene, cd, 5, prec, 0, H, buf.data_ptr(), buf.numel() * buf.element_size(), None)


for n_threads in (1, 2, 4):
    ctxs = [rta.Context(0) for _ in range(n_threads)]
    scs = [rta.Scene(c, flat) for c in ctxs]
    bufs = [torch.empty((H, W, 3), dtype=dtype, device="cuda") for _ in range(n_threads)]
    for c, s, b in zip(ctxs, scs, bufs):
        worker(c, s, b, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = [threading.Thread(target=worker, args=(c, s, b, frames)) for c, s, b in zip(ctxs, scs, bufs)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s %s: %d thread(s): %.2f ms per frame of wall time, %.1f frames/s" % (name, "f32" if prec == rta.F32 else "f64", n_threads,
                                                                           dt * 1e3 / (frames * n_threads), frames * n_threads / dt))
    for s in scs:
        s.close()
    for c in ctxs:
        c.close()
