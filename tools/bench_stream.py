"""BASELINE configs[4] on one GPU: synthetic 1080p frames with K in [1,16] face boxes each (sides 96..400 px)
-> square-box maths + crop + bilinear resize to 256x256 (prediction.py:36-82) -> batched landmarks -> similarity +
alignment warp to 256x256.  Frames are resident in HBM (a decoder would put them there); reports frames/s and
faces/s for one frame per launch sequence and for several frames batched together.

    python tools/bench_stream.py            # fp32;  DTYPE=bf16 for the bf16 configuration
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
from flm_amd import alignment, prediction
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights

dtype = os.environ.get("DTYPE", "f32")
NF = int(os.environ.get("FRAMES", "64"))
rng = np.random.default_rng(5)
model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dtype)
model.load_weights(synth_fcn8_weights(68, 2))
frames = [torch.from_numpy(rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)).cuda() for _ in range(8)]
faces = []
for _ in range(NF):
    k = int(rng.integers(1, 17))
    fb = []
    for _ in range(k):
        side = int(rng.integers(96, 401))
        x0 = int(rng.integers(0, 1920 - side)); y0 = int(rng.integers(0, 1080 - side))
        fb.append((x0, y0, x0 + side, y0 + side))
    faces.append(fb)
tmpl = torch.from_numpy(alignment.canonical_template(68, 256, 256)).cuda()
scale = (256 / 264, 256 / 264)


def run(group):
    """`group` frames per launch sequence."""
    n_faces = 0
    for f0 in range(0, NF, group):
        crops = []
        for f in range(f0, min(f0 + group, NF)):
            boxes = prediction.face_boxes(faces[f])
            crops.append(prediction.crop_faces_device(frames[f % len(frames)], boxes, 256, 256))
        crops = torch.cat(crops, 0) if len(crops) > 1 else crops[0]
        lm = model.forward_device(crops, "landmarks", n_points=4)
        alignment.align_device(crops, lm, tmpl, 256, 256, scale)
        n_faces += crops.shape[0]
    return n_faces


for group in (1, 4, 16):
    run(group)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nf = run(group)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s, %2d frame(s) per batch: %7.1f frames/s, %8.1f faces/s (%d frames, %d faces, %.1f faces per batch)"
          % (dtype, group, NF / dt, nf / dt, NF, nf, nf / (NF / group)), flush=True)
