"""Developer tool: a few small-batch forwards for rocprofv3 --kernel-trace --stats (B and DTYPE from the environment)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights
B = int(os.environ.get("B", "1"))
dtype = os.environ.get("DTYPE", "f32")
m = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dtype)
m.load_weights(synth_fcn8_weights(68, 2))
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
for _ in range(20):
    m.forward_device(x, "landmarks", n_points=4)
torch.cuda.synchronize()
print("done")
