"""Developer tool: bf16 batch-512 forward + selection, per-layer times with the 256-row tiles filled by LDS-DMA vs
through staging registers (flm_set_tuning "bf16_lds_dma")."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
from flm_amd import _lib
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights

B = int(os.environ.get("B", "512"))
lib = _lib.load()
model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype="bf16")
model.load_weights(synth_fcn8_weights(68, 2))
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
for rep in range(2):
    for dma in (0, 1):
        _lib.check(lib.flm_set_tuning(b"bf16_lds_dma", dma), "set_tuning")
        for _ in range(3):
            model.forward_device(x, "landmarks", n_points=4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model.forward_device(x, "landmarks", n_points=4)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("lds_dma %d: %.3f ms per batch of %d  (%.0f faces/s)" % (dma, 1e3 * dt, B, B / dt), flush=True)
