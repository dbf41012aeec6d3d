"""Developer tool: per-layer launch durations (HIP events) of forward + landmark selection at small batches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
import bench
from flm_amd import _lib
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights

lib = _lib.load()
w = synth_fcn8_weights(68, 2)
for dtype in ("f32", "bf16"):
    m = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dtype)
    m.load_weights(w)
    for B in [int(v) for v in os.environ.get("BS", "1,4,16").split(",")]:
        x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
        for _ in range(5):
            m.forward_device(x, "landmarks", n_points=4)
        torch.cuda.synchronize()
        _lib.check(lib.flm_profile_enable(4096), "flm_profile_enable")
        lib.flm_profile_filter(None)
        lib.flm_profile_reset()
        for _ in range(20):
            m.forward_device(x, "landmarks", n_points=4)
        torch.cuda.synchronize()
        layers = bench.read_profile(lib)
        lib.flm_profile_disable()
        print(dtype, "batch", B, "sum %.3f ms:" % sum(layers.values()),
              " ".join("%s %.0f" % (k, 1e3 * v) for k, v in layers.items()), "(us)", flush=True)
