import ctypes, os, sys
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
import torch, numpy as np
rta = load_package()
from ray_tracer_amd import scenes
W, H = 3840, 2160
w, cam = scenes.SCENES["random"](rta.api, hsize=W, vsize=H)
lib = ctypes.CDLL(os.environ["RTX_LIB"])
ctx = rta.Context(0); scene = rta.Scene(ctx, rta.flatten(w)); cd = rta.camera_desc(cam)
tile = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
rta.render_device(ctx, scene, cd, 5, rta.F32, 0, H, tile.data_ptr(), tile.numel()*4, None, {})
out = np.zeros(64*8)
assert lib.rtx_debug_slow_rays_f32(out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
np.set_printoptions(linewidth=200, precision=6, suppress=False)
print(out.reshape(64,8)[:, :8])
