"""Developer tool: a few forwards with landmark selection, for use under rocprofv3 (kernel trace / PMC passes).
B (512), DTYPE (bf16), STEPS (3), KNOBS (flm_set_tuning key=value pairs) from the environment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
from flm_amd import _lib
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights

for kv in filter(None, os.environ.get("KNOBS", "").split(",")):
    k, v = kv.split("=")
    _lib.check(_lib.load().flm_set_tuning(k.encode(), int(v)), "set_tuning")
B = int(os.environ.get("B", "512"))
dtype = os.environ.get("DTYPE", "bf16")
model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dtype)
model.load_weights(synth_fcn8_weights(68, 2))
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
for _ in range(int(os.environ.get("STEPS", "3"))):
    model.forward_device(x, "landmarks", n_points=4)
torch.cuda.synchronize()
print("done")
