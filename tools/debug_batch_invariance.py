"""Developer tool: which layer's bits depend on the batch a face sits in (fp32).  Compares the intermediates of the
first faces between two batch sizes; neither run imports oracle/."""
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
for kv in filter(None, os.environ.get("KNOBS", "").split(",")):
    k, v = kv.split("=")
    _lib.check(lib.flm_set_tuning(k.encode(), int(v)), "set_tuning")
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64,512").split(",")]
nmax = max(sizes)
crops = np.random.default_rng(7).integers(0, 256, (nmax, 256, 256, 3), dtype=np.uint8)
model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256)
model.load_weights(synth_fcn8_weights(68, 2))
xd = torch.from_numpy(crops).cuda()
names = ("f1", "f2", "f3", "f4", "f5", "fc6", "fc7", "fuse4", "seg_feats")
ref = None
for n in sizes:
    model._ws.clear()
    pr = model.forward_device(xd[:n].contiguous(), "probs")[:4].cpu().numpy()
    torch.cuda.synchronize()
    cur = {k: model.intermediate(k, n, "probs")[:4].cpu().numpy() for k in names}
    cur["probs"] = pr
    if ref is None:
        ref = cur
        continue
    for k in list(names) + ["probs"]:
        d = np.abs(cur[k] - ref[k])
        print("batch %4d vs %4d  %-9s equal=%s  max |diff| %.3g  differing %.4f %%" %
              (n, sizes[0], k, np.array_equal(cur[k], ref[k]), d.max(), 100 * (d > 0).mean()))
