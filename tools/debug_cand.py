"""Developer tool: candidate-path bookkeeping of one forward (list fill, overflow flag, thresholds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
from flm_amd import _lib
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights

lib = _lib.load()
B = int(os.environ.get("B", "4")); NP = int(os.environ.get("NP", "4"))
model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=os.environ.get("DTYPE", "f32"))
model.load_weights(synth_fcn8_weights(68, 2))
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
ws = model.new_workspace(B, "landmarks", NP)
lm = model.forward_device(x, "landmarks", n_points=NP, workspace=ws)
torch.cuda.synchronize()
def off(name):
    return lib.flm_fcn8_workspace_offset(name.encode(), B, 256, 256, 68, model._dt, _lib.OUT_LANDMARKS, 1, NP)
cap = off("cand_cap")
cnt = ws[off("cand_cnt"):off("cand_cnt") + 4 * (B + 1)].view(torch.int32).cpu().numpy()
tau = ws[off("cand_tau"):off("cand_tau") + 4 * B * 68].view(torch.float32).view(B, 68).cpu().numpy()
print("cap", cap, "counts", cnt[:B], "overflow", cnt[B])
print("tau[0][:8]", tau[0][:8])
probs = model.forward_device(x, "probs").cpu().numpy().reshape(B, 264 * 264, 68)
print("pixels >= tau per class, face 0:", (probs[0] >= tau[0][None]).sum(0)[:16], "total", (probs[0] >= tau[0][None]).sum())

