"""Developer tool: does alternating two HIP streams between consecutive steps (step i's under-filled tail kernels
overlap step i+1's first layers) raise the throughput of the bench step?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
from flm_amd import alignment
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights

B = int(os.environ.get("B", "64")); dtype = os.environ.get("DTYPE", "f32"); STEPS = int(os.environ.get("STEPS", "20"))
w = synth_fcn8_weights(68, 2)
models = []
for _ in range(2):
    m = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dtype)
    m.load_weights(w)
    models.append(m)
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
tmpl = torch.from_numpy(alignment.canonical_template(68, 256, 256)).cuda()
scale = (256 / 264, 256 / 264)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def step(i, nstreams):
    k = i % nstreams
    with torch.cuda.stream(streams[k]):
        lm = models[k].forward_device(x, "landmarks", n_points=4)
        alignment.align_device(x, lm, tmpl, 256, 256, scale)

for ns in (1, 2, 1, 2):
    for i in range(4):
        step(i, ns)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        step(i, ns)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("streams %d: %.3f ms/step, %.0f faces/s" % (ns, 1e3 * dt / STEPS, B * STEPS / dt))
