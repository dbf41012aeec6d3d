"""Developer tool: sweep a tuning knob of libflm_hip.so and print per-layer times (HIP events)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
from flm_amd import _lib
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights

lib = _lib.load()
B = int(os.environ.get("B", "64"))
model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=os.environ.get("DTYPE", "f32"))
model.load_weights(synth_fcn8_weights(68, 2))
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()


def run(steps=6):
    for _ in range(2):
        model.forward_device(x, "landmarks", n_points=4)
    torch.cuda.synchronize()
    lib.flm_profile_enable(4096)
    lib.flm_profile_reset()
    for _ in range(steps):
        model.forward_device(x, "landmarks", n_points=4)
    torch.cuda.synchronize()
    ms = {}
    name = C.create_string_buffer(32); v = C.c_float(); i = 0
    while lib.flm_profile_read(i, name, 32, C.byref(v)) == 0:
        ms.setdefault(name.value.decode(), []).append(v.value); i += 1
    lib.flm_profile_disable()
    return {k: float(np.median(a)) for k, a in ms.items()}


key = sys.argv[1]
vals = [int(v) for v in sys.argv[2].split(",")]
for v in vals:
    _lib.check(lib.flm_set_tuning(key.encode(), v), "set_tuning")
    r = run()
    keys = list(r.keys())
    print("%-8s" % key[:8], " ".join("%7s" % k[:7] for k in keys), "   total")
    print("%-8d" % v, " ".join("%7.3f" % r[k] for k in keys), "  %7.3f" % sum(r.values()))
