"""Developer tool: run one packed conv layer on the GPU for several shapes and compare with
torch's own conv on the same device (quick localisation of kernel bugs; not a parity test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
import flm_amd
from flm_amd import _lib
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights

lib = _lib.load()
params = synth_fcn8_weights(68, 2)
model = LANDMARKS_MODELS["fcn_8"](68, input_height=32, input_width=32)
model.load_weights(params)
layer = sys.argv[1] if len(sys.argv) > 1 else "fc6"
cin = {"fc6": 256, "fc7": 4096, "enc2": 64, "enc3": 128, "enc4": 256, "enc5": 256}[layer]
w = torch.from_numpy(params[layer + "/kernel"]).cuda().permute(3, 2, 0, 1).contiguous()
b = torch.from_numpy(params[layer + "/bias"]).cuda()
pad = w.shape[2] // 2
for (n, h, ww) in [(2, 2, 3), (2, 4, 4), (1, 8, 8), (2, 8, 8), (3, 8, 8), (64, 8, 8), (2, 8, 4), (2, 5, 8)]:
    x = torch.randn(n, h, ww, cin, device="cuda")
    y = torch.empty(n, h, ww, w.shape[0], device="cuda")
    _lib.check(lib.flm_fcn8_run_layer(_lib.stream_ptr(), _lib.ptr(model._packed), layer.encode(), _lib.ptr(x),
                                      _lib.ptr(y), n, h, ww, 68, 0), "run_layer")
    ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=pad)).permute(0, 2, 3, 1)
    err = (y - ref).abs()
    bad = (err > 1e-3 * ref.abs().max()).nonzero()
    print(layer, (n, h, ww), "max err", float(err.max()), "ref max", float(ref.abs().max()), "nbad", len(bad),
          "first bad", bad[:3].tolist() if len(bad) else "")
    if len(bad):
        pos = torch.unique(bad[:, 0] * 1000 + bad[:, 1] * 10 + bad[:, 2])
        print("   bad (n,y,x) count", len(pos), pos[:20].tolist())
