"""Developer tool: per-launch HIP-event times of one registry model (MODEL, DTYPE, B env)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
from flm_amd import _lib
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd import weights as W
lib = _lib.load()
name = os.environ.get("MODEL", "fcn_8_vgg"); dtype = os.environ.get("DTYPE", "f32"); B = int(os.environ.get("B", "64"))
mk = {"vgg": W.synth_vgg_weights, "mobilenet": W.synth_mobilenet_weights, "resnet50": W.synth_resnet50_weights}
key = [k for k in mk if k in name]
wts = mk[key[0]](68, 2, fcn32="fcn_32" in name) if key else (W.synth_fcn32_weights(68, 2) if "fcn_32" in name else W.synth_fcn8_weights(68, 2))
m = LANDMARKS_MODELS[name](68, input_height=256, input_width=256, dtype=dtype)
m.load_weights(wts)
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
for _ in range(2):
    m.forward_device(x, "landmarks", n_points=4)
torch.cuda.synchronize()
lib.flm_profile_enable(4096); lib.flm_profile_reset()
m.forward_device(x, "landmarks", n_points=4)
torch.cuda.synchronize()
nm = C.create_string_buffer(32); v = C.c_float(); i = 0; tot = 0
while lib.flm_profile_read(i, nm, 32, C.byref(v)) == 0:
    print("%3d %-16s %8.3f" % (i, nm.value.decode(), v.value)); tot += v.value; i += 1
print("total %.3f ms" % tot)
