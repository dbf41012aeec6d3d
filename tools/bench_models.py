"""Developer tool: forward + top-4 landmark selection of every registry model (synthetic weights), ms per batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd import weights as W

B = int(os.environ.get("B", "64"))
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
synth = {"fcn_8": lambda: W.synth_fcn8_weights(68, 2), "fcn_32": lambda: W.synth_fcn32_weights(68, 2),
         "fcn_8_vgg": lambda: W.synth_vgg_weights(68, 2), "fcn_32_vgg": lambda: W.synth_vgg_weights(68, 2, fcn32=True),
         "fcn_8_mobilenet": lambda: W.synth_mobilenet_weights(68, 2), "fcn_32_mobilenet": lambda: W.synth_mobilenet_weights(68, 2, fcn32=True),
         "fcn_8_resnet50": lambda: W.synth_resnet50_weights(68, 2), "fcn_32_resnet50": lambda: W.synth_resnet50_weights(68, 2, fcn32=True)}
for name, mk in synth.items():
    for dtype in ("f32", "bf16"):
        m = LANDMARKS_MODELS[name](68, input_height=256, input_width=256, dtype=dtype)
        m.load_weights(mk())
        for _ in range(2):
            m.forward_device(x, "landmarks", n_points=4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m.forward_device(x, "landmarks", n_points=4)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("%-18s %-4s batch %d: %7.2f ms  %8.0f faces/s" % (name, dtype, B, 1e3 * dt, B / dt), flush=True)
        del m
        torch.cuda.empty_cache()
