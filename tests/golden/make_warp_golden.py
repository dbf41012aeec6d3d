"""Generate tests/golden/warp_golden.npz with scikit-image 0.18.3 (third-party library, not
reference code) -- the library call shape the reference's only affine warp uses
(data/generator.py:192-200: `transform.warp(im, tform, mode="edge")`).

Run in the build container with the second interpreter:
    /opt/conda/bin/python3.9 tests/golden/make_warp_golden.py
"""
import os

import numpy as np
from skimage import transform

rng = np.random.default_rng(42)
out = {}
n = 4
hs, ws, hd, wd = 48, 56, 40, 44
imgs = rng.integers(0, 256, (n, hs, ws, 3), dtype=np.uint8)
# smooth the images a little so that sub-pixel differences do not dominate
imgs = ((imgs.astype(np.float64) + np.roll(imgs, 1, 1) + np.roll(imgs, 1, 2)) / 3).astype(np.uint8)
k = 12
src_pts = rng.uniform(5, 40, (n, k, 2))
mats = np.zeros((n, 3, 3))
warped = np.zeros((n, hd, wd, 3))
dst_pts = np.zeros((n, k, 2))
for i in range(n):
    s = rng.uniform(0.6, 1.5)
    th = rng.uniform(-0.6, 0.6)
    t = rng.uniform(-6, 6, 2)
    R = s * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    dst = src_pts[i] @ R.T + t + rng.normal(0, 0.3, (k, 2))   # noisy similarity
    dst_pts[i] = dst
    tf = transform.SimilarityTransform()
    assert tf.estimate(src_pts[i], dst)
    mats[i] = tf.params
    warped[i] = transform.warp(imgs[i], tf.inverse, output_shape=(hd, wd), order=1, mode="edge",
                               preserve_range=True)
out.update(imgs=imgs, src_pts=src_pts, dst_pts=dst_pts, mats=mats, warped=warped)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "warp_golden.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst))
