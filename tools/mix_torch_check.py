import sys, os
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
rta = load_package()
from ray_tracer_amd import scenes
import numpy as np
w, cam = scenes.default_scene(rta.api, 64, 48)
a = rta.render_rows(cam, w)
import torch
print("torch sees", torch.cuda.device_count(), torch.cuda.is_available())
x = torch.ones(4, device="cuda") * 2
print(x.sum().item(), a.sum())
print(sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}))
