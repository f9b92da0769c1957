"""Does one frame finish sooner when its rows are rendered as several parts side by side on separate
streams?  The bounce levels of a scene with the sphere hierarchy leave most of the chip idle; the
camera launch of another part can use it.  Renders the same 4K frame whole, then as 2, 3 and 4 row
ranges from as many host threads (one context each, one buffer), and prints milliseconds per frame.
python3 tools/split_frame.py [scene] [f64|f32] [frames]   (on a GPU box)"""
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
buf = torch.empty((H, W, 3), dtype=dtype, device="cuda")
row_bytes = W * 3 * buf.element_size()
ref = None
for n_parts in (1, 2, 3, 4, 1):
    ctxs = [rta.Context(0) for _ in range(n_parts)]
    scs = [rta.Scene(c, flat) for c in ctxs]
    cuts = [(H * k // n_parts) // 8 * 8 for k in range(n_parts)] + [H]
    barrier = threading.Barrier(n_parts + 1)

    def worker(k):
        a, b = cuts[k], cuts[k + 1]
        for _ in range(frames + 3):
            barrier.wait()
            rta.render_device(ctxs[k], scs[k], cd, 5, prec, a, b, buf.data_ptr() + a * row_bytes, (b - a) * row_bytes, None)
            barrier.wait()

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(n_parts)]
    for t in ts:
        t.start()
    times = []
    for f in range(frames + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        barrier.wait()
        barrier.wait()  # every part's call has returned (a call without a stream returns a finished part)
        times.append((time.perf_counter() - t0) * 1e3)
    for t in ts:
        t.join()
    s = float(buf.double().sum())
    if ref is None:
        ref = s
    times = sorted(times[3:])
    print("%s %s: %d part(s): median %.3f ms, best %.3f ms per frame; checksum %s" %
          (name, "f32" if prec == rta.F32 else "f64", n_parts, times[len(times) // 2], times[0], "same" if abs(s - ref) <= 1e-6 * abs(ref) else "DIFFERENT"), flush=True)
    for sc in scs:
        sc.close()
    for c in ctxs:
        c.close()
