"""Generate tests/golden/decode_golden.npz by RUNNING the reference's own decode.

Run in the build container only (the reference never travels):
    MPLBACKEND=Agg python tests/golden/make_decode_golden.py

Imports /root/reference/keypoints_detector/utils/metrics.py (numpy + matplotlib
only, no TensorFlow) and records, for seeded heatmaps, the outputs of
`get_average_xy` (utils/metrics.py:46-80) and of `transfer_target` as shipped
(:102-109).  The fixture holds inputs and expected outputs only -- data, no
reference source.
"""
import importlib.util
import os
import sys

import numpy as np

REF = "/root/reference/keypoints_detector/utils/metrics.py"
spec = importlib.util.spec_from_file_location("ref_metrics", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(20261004)
out = {}


def gauss(h, w, cx, cy, s):
    y, x = np.mgrid[0:h, 0:w]
    return np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * s * s)).astype(np.float32)


cases = []
# (name, heatmap) -- continuous random values so that argsort ties do not occur
cases.append(("rand96", rng.random((96, 96), dtype=np.float32)))
cases.append(("rand264", rng.random((264, 264), dtype=np.float32)))
cases.append(("rand_rect", rng.random((40, 72), dtype=np.float32)))
cases.append(("gauss256", gauss(256, 256, 100.3, 57.8, 3.0)))
cases.append(("gauss264_corner", gauss(264, 264, 1.2, 261.7, 2.0)))
cases.append(("gauss96_noise", (gauss(96, 96, 40.5, 60.25, 4.0) + 0.01 * rng.random((96, 96), dtype=np.float32)).astype(np.float32)))
cases.append(("zeros96", np.zeros((96, 96), np.float32)))
cases.append(("softmaxlike", (lambda z: (np.exp(z) / np.exp(z).sum()).astype(np.float32))(rng.normal(size=(64, 64)) * 3)))

modes = [0, 1, 4, 9, 25, 64]
thresholds = [0, 0.2]
names = []
for name, hm in cases:
    names.append(name)
    out["hm_" + name] = hm
    res = np.zeros((len(modes), len(thresholds), 2), np.float64)
    for i, n in enumerate(modes):
        for j, t in enumerate(thresholds):
            with np.errstate(all="ignore"):
                xy = ref.get_average_xy(hm, hm.shape[0], hm.shape[1], n, t)
            res[i, j] = [float(xy[0]), float(xy[1])]
    out["xy_" + name] = res

# transfer_target exactly as shipped (positional slip -> n=4, thresh=0)
y_pred = rng.random((3, 48, 56, 5), dtype=np.float32)
y_pred[1, :, :, 2] = 0.0  # an all-zero landmark map -> (-1,-1)
out["tt_input"] = y_pred
with np.errstate(all="ignore"):
    out["tt_shipped_default"] = np.asarray(ref.transfer_target(y_pred), np.float64)
    out["tt_shipped_args"] = np.asarray(ref.transfer_target(y_pred, 0.2, 25), np.float64)
out["modes"] = np.array(modes)
out["thresholds"] = np.array(thresholds, np.float64)
out["names"] = np.array(names)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_golden.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes")
print("gauss256 rows (n=0,1,4,9,25,64 ; thresh=0):")
print(out["xy_gauss256"][:, 0])
