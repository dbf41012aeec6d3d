#!/usr/bin/env python3
"""Developer tool: A/B of the MFMA shape of the bf16 256x256-tile kernel (flm_set_tuning "bf16_mfma16"), same process,
alternating: per-layer launch durations of the batch-512 forward."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
import bench
from flm_amd import _lib
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights

lib = _lib.load()
B = int(os.environ.get("B", "512"))
model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype="bf16")
model.load_weights(synth_fcn8_weights(68, 2))
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
for rnd in range(3):
    for knob in (0, 1):
        _lib.check(lib.flm_set_tuning(os.environ.get("KNOB", "bf16_mfma16").encode(), knob), "set_tuning")
        for _ in range(3):
            model.forward_device(x, "landmarks", n_points=4)
        torch.cuda.synchronize()
        _lib.check(lib.flm_profile_enable(4096), "flm_profile_enable")
        lib.flm_profile_filter(None)
        lib.flm_profile_reset()
        for _ in range(10):
            model.forward_device(x, "landmarks", n_points=4)
        torch.cuda.synchronize()
        layers = bench.read_profile(lib)
        lib.flm_profile_disable()
        print("mfma16=%d: sum %.3f ms " % (knob, sum(layers.values())) +
              " ".join("%s %.3f" % (k, layers[k]) for k in ("enc2", "enc3", "enc4", "enc5", "fc6", "fc7")), flush=True)
_lib.check(lib.flm_set_tuning(os.environ.get("KNOB", "bf16_mfma16").encode(), 1), "set_tuning")
