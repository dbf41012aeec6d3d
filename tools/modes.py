"""Developer tool: the three output modes of the forward at the headline batch (probabilities as the reference's
model.predict returns them, the class map of _prediction, landmarks), ms per batch on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights
w = synth_fcn8_weights(68, 2)
for dtype, B in (("f32", 64), ("bf16", 512)):
    m = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dtype)
    m.load_weights(w)
    x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
    for mode, kw in (("probs", {}), ("classmap", {}), ("landmarks", {"n_points": 4}), ("landmarks", {"n_points": 0})):
        for _ in range(3):
            m.forward_device(x, mode, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            m.forward_device(x, mode, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("%s batch %d %-9s %s: %.3f ms  %.0f faces/s" % (dtype, B, mode, kw, 1e3 * dt, B / dt), flush=True)
    del m
    torch.cuda.empty_cache()
