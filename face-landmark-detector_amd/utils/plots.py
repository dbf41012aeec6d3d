"""Host-side visualisation after the path (reference utils/plots.py:60-72,116-166), without cv2.

Only what `keypts_predict(..., out_fname=...)` needs: colourise a class map and overlay it.
"""
from __future__ import annotations

import random

import numpy as np

random.seed(0)
class_colors = [(random.randint(0, 255), random.randint(0, 255), random.randint(0, 255)) for _ in range(5000)]


def get_colored_segmentation_image(seg_arr, n_classes, colors=None):
    """utils/plots.py:60-72."""
    colors = class_colors if colors is None else colors
    out = np.zeros(seg_arr.shape + (3,), np.float64)
    for c in range(n_classes):
        m = (seg_arr == c)
        for ch in range(3):
            out[:, :, ch] += m * colors[c][ch]
    return out


def _resize_nearest(img, w, h):
    ys = (np.arange(h) * img.shape[0] / h).astype(np.int64)
    xs = (np.arange(w) * img.shape[1] / w).astype(np.int64)
    return img[ys][:, xs]


def visualize_keypoints(seg_arr, inp_img=None, n_classes=None, colors=None, class_names=None, overlay_img=False,
                        show_legends=False, pred_dim=None):
    """utils/plots.py:116-149 (legends omitted)."""
    if n_classes is None:
        n_classes = int(np.max(seg_arr)) + 1
    seg_img = get_colored_segmentation_image(seg_arr, n_classes, colors=colors)
    if inp_img is not None:
        seg_img = _resize_nearest(seg_img, inp_img.shape[1], inp_img.shape[0])
    if pred_dim is not None:
        seg_img = _resize_nearest(seg_img, pred_dim[1], pred_dim[0])
        if inp_img is not None:
            inp_img = _resize_nearest(inp_img, pred_dim[1], pred_dim[0])
    if overlay_img and inp_img is not None:
        seg_img = (inp_img / 2 + seg_img / 2)
    return seg_img.astype(np.uint8)


def draw_marks(image, marks, color=(0, 255, 0)):
    """utils/plots.py:152-166: mark each landmark (2x2 dot; cv2.circle is not available)."""
    for mark in np.asarray(marks).reshape(-1, 2):
        x, y = int(mark[0]), int(mark[1])
        image[max(y - 1, 0):y + 1, max(x - 1, 0):x + 1] = color
    return image
