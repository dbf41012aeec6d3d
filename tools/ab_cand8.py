#!/usr/bin/env python3
"""Developer tool: A/B of the bf16 candidate launch of up3 (flm_set_tuning "up3_cand8": 8-wave kernel vs generic kernel),
batch 512, per-layer HIP-event times and bit-equality of the landmarks."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flm_amd  # noqa: E402,F401
from flm_amd import _lib  # noqa: E402
from flm_amd.networks import LANDMARKS_MODELS  # noqa: E402
from flm_amd.weights import synth_fcn8_weights  # noqa: E402

lib = _lib.load()
B = int(os.environ.get("B", "512"))
model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype="bf16")
model.load_weights(synth_fcn8_weights(68, 2))
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
out = {}
ROWS = [int(v) for v in os.environ.get("ROWS", "0").split(",")]
KNOBS = [int(v) for v in os.environ.get("KNOBS", "1").split(",")]   # up3_cand8 values: 1 = 8 waves, 5 = 4 waves x 2 workgroups
for npts in (4, 25):
    for knob, rows in ([(0, 0)] + [(k, r) for k in KNOBS for r in ROWS]) * 2:
        _lib.check(lib.flm_set_tuning(b"up3_cand8", knob), "set_tuning")
        _lib.check(lib.flm_set_tuning(b"up3_cand8_rows", rows), "set_tuning")
        for _ in range(3):
            lm = model.forward_device(x, "landmarks", n_points=npts)
        torch.cuda.synchronize()
        lib.flm_profile_enable(400); lib.flm_profile_filter(None); lib.flm_profile_reset()
        for _ in range(5):
            lm = model.forward_device(x, "landmarks", n_points=npts)
        torch.cuda.synchronize()
        ms = {}
        name = C.create_string_buffer(32); v = C.c_float(); i = 0
        while lib.flm_profile_read(i, name, 32, C.byref(v)) == 0:
            ms.setdefault(name.value.decode(), []).append(v.value); i += 1
        lib.flm_profile_disable()
        out[(npts, min(knob, 1))] = lm.cpu().numpy()
        print("n_points %2d cand8=%d rows=%d: up3 %.3f ms  up3_sub %.3f  decode %.3f  fallback %.3f  total %.3f" %
              (npts, knob, rows, np.median(ms["up3"]), np.median(ms["up3_sub"]), np.median(ms["decode"]),
               np.median(ms.get("up3_fallback", [0])), sum(float(np.median(a)) for a in ms.values())), flush=True)
    print("n_points %d: landmarks equal between the two kernels: %s" % (npts, np.array_equal(out[(npts, 0)], out[(npts, 1)])))
