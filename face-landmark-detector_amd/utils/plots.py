"""Host-side visualisation after the path (reference utils/plots.py:60-166), without cv2.

Mirrors `get_colored_segmentation_image` (:60-72), `get_legends` (:75-91), `concat_lenends` (:94-104, the
reference's spelling), `overlay_seg_image` (:107-114), `visualize_keypoints` (:116-149) and `draw_marks`
(:152-166) on numpy / PIL.  Arrays are BGR uint8 as cv2 would hold them.  Where the reference leans on cv2:

* `cv2.resize(..., INTER_NEAREST)` -> index arithmetic `src = floor(dst * src_size / dst_size)` (cv2's rule);
* `cv2.resize(inp_img, ...)` (INTER_LINEAR, uint8) -> the 11-bit fixed-point algorithm of OpenCV's generic
  8-bit path, the same specification the device kernel `flm_crop_resize` implements (csrc/flm_misc.hip);
* `cv2.rectangle(..., -1)` -> filled box, both corners inclusive, clipped to the image;
* `cv2.putText(FONT_HERSHEY_COMPLEX, 0.5)` -> PIL's built-in bitmap font at the same anchor (the glyph shapes
  differ from Hershey's; everything else about the legend -- size, swatches, background -- is the reference's);
* `cv2.circle(img, (x, y), 2, color, -1, cv2.LINE_AA)` -> the 21-pixel filled disc cv2 rasterises for radius 2
  (rows of 3, 5, 5, 5, 3 pixels); LINE_AA's blended rim is not reproduced.
Defects of the reference routed around: `pred_dim=None` no longer raises (:125 unpacks it unconditionally) and
`n_classes=None` means max+1 (:127 drops the last class).
"""
from __future__ import annotations

import random

import numpy as np

random.seed(0)
class_colors = [(random.randint(0, 255), random.randint(0, 255), random.randint(0, 255)) for _ in range(5000)]


def get_colored_segmentation_image(seg_arr, n_classes, colors=None):
    """utils/plots.py:60-72: float64 [H,W,3], channel k of a pixel = colors[class][k]."""
    colors = class_colors if colors is None else colors
    seg_arr = np.asarray(seg_arr)
    out = np.zeros(seg_arr.shape[:2] + (3,), np.float64)
    for c in range(n_classes):
        m = (seg_arr[:, :] == c)
        for ch in range(3):
            out[:, :, ch] += (m * colors[c][ch]).astype("uint8")
    return out


def _resize_nearest(img, w, h):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_NEAREST): src index = floor(dst * src / dst_size)."""
    ys = np.minimum((np.arange(h) * (img.shape[0] / h)).astype(np.int64), img.shape[0] - 1)
    xs = np.minimum((np.arange(w) * (img.shape[1] / w)).astype(np.int64), img.shape[1] - 1)
    return img[ys][:, xs]


def _linear_coef(n_dst, n_src, along_y=False):
    """Tap indices and 11-bit weights of OpenCV's 8-bit INTER_LINEAR.  Along x a position outside [0, n_src - 1] folds
    onto the border pixel with weights 2048 | 0; along y only the row indices clamp and the split weights stay."""
    scale = np.float64(1.0) / (np.float64(n_dst) / np.float64(n_src))
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if along_y:
        s0, s1 = np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1)
    else:
        low, high = s < 0, s >= n_src - 1
        s0 = np.where(low, 0, np.where(high, n_src - 1, s))
        s1 = np.minimum(s0 + 1, n_src - 1)
        f = np.where(low | high, np.float32(0), f).astype(np.float32)
    return (s0, s1, np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64),
            np.rint(f * np.float32(2048)).astype(np.int64))


def _resize_linear_u8(img, w, h):
    """cv2.resize(img, (w, h)) for uint8 (default INTER_LINEAR): OpenCV's 11-bit fixed-point weights, the exact-2x
    case averaged as INTER_AREA -- the specification of csrc/flm_misc.hip's crop_resize_kernel, on the host."""
    img = np.asarray(img)
    if img.dtype != np.uint8:
        return _resize_nearest(img, w, h)
    s = np.atleast_3d(img).astype(np.int64)
    sh, sw = s.shape[:2]
    if (sh, sw) == (h, w):
        return img.copy()
    if sw == 2 * w and sh == 2 * h:
        out = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    else:
        x0, x1, a0, a1 = _linear_coef(w, sw)
        y0, y1, b0, b1 = _linear_coef(h, sh, along_y=True)
        hrow = s[:, x0] * a0[None, :, None] + s[:, x1] * a1[None, :, None]
        out = (((b0[:, None, None] * (hrow[y0] >> 4)) >> 16) + ((b1[:, None, None] * (hrow[y1] >> 4)) >> 16) + 2) >> 2
    out = out.astype(np.uint8)
    return out if img.ndim == 3 else out[:, :, 0]


def get_legends(class_names, colors=None):
    """utils/plots.py:75-91: white uint8 [(25*len + 25), 125, 3]; per class its name at (5, 25*i + 17) in black and a
    filled swatch over columns 100..124, rows 25*i..25*i+25 (cv2.rectangle's corners are inclusive)."""
    from PIL import Image, ImageDraw
    colors = class_colors if colors is None else colors
    n_classes = len(class_names)
    legend = np.zeros(((len(class_names) * 25) + 25, 125, 3), dtype="uint8") + 255
    img = Image.fromarray(legend)
    draw = ImageDraw.Draw(img)
    for i, (class_name, color) in enumerate(zip(class_names[:n_classes], colors[:n_classes])):
        color = tuple(int(c) for c in color)
        # cv2.putText's origin is the text's bottom-left corner; PIL's the top-left: 11 px up for the ~11 px glyphs
        draw.text((5, (i * 25) + 17 - 11), str(class_name), fill=(0, 0, 0))
        draw.rectangle([100, i * 25, 125, (i * 25) + 25], fill=color)
    return np.asarray(img).copy()


def concat_lenends(seg_img, legend_img):
    """utils/plots.py:94-104: legend on the left, picture on the right, background = the legend's first value."""
    new_h = np.maximum(seg_img.shape[0], legend_img.shape[0])
    new_w = seg_img.shape[1] + legend_img.shape[1]
    out_img = np.zeros((new_h, new_w, 3)).astype("uint8") + legend_img[0, 0, 0]
    out_img[:legend_img.shape[0], :legend_img.shape[1]] = np.copy(legend_img)
    out_img[:seg_img.shape[0], legend_img.shape[1]:] = np.copy(seg_img)
    return out_img


def overlay_seg_image(inp_img, seg_img):
    """utils/plots.py:107-114."""
    seg_img = _resize_nearest(seg_img, inp_img.shape[1], inp_img.shape[0])
    return (inp_img / 2 + seg_img / 2).astype("uint8")


def visualize_keypoints(kpts_arr, inp_img=None, n_classes=None, colors=None, class_names=None, overlay_img=False,
                        show_legends=False, pred_dim=None):
    """utils/plots.py:116-149.  `pred_dim` = (prediction_width, prediction_height) as there."""
    prediction_width, prediction_height = pred_dim if pred_dim is not None else (None, None)
    kpts_arr = np.asarray(kpts_arr)
    if n_classes is None:
        n_classes = int(np.max(kpts_arr)) + 1
    seg_img = get_colored_segmentation_image(kpts_arr, n_classes, colors=colors)
    if inp_img is not None:
        seg_img = _resize_nearest(seg_img, inp_img.shape[1], inp_img.shape[0])
    if (prediction_height is not None) and (prediction_width is not None):
        seg_img = _resize_nearest(seg_img, prediction_width, prediction_height)
        if inp_img is not None:
            inp_img = _resize_linear_u8(inp_img, prediction_width, prediction_height)
    if overlay_img:
        assert inp_img is not None
        seg_img = overlay_seg_image(inp_img, seg_img)
    if show_legends:
        assert class_names is not None
        legend_img = get_legends(class_names, colors=colors)
        seg_img = concat_lenends(seg_img, legend_img)
    return seg_img


# cv2.circle(img, c, 2, color, -1): the filled disc of radius 2 is rows of 3, 5, 5, 5, 3 pixels (dx^2 + dy^2 <= 5)
_DISC2 = [(dx, dy) for dy in range(-2, 3) for dx in range(-2, 3) if dx * dx + dy * dy <= 5]


def draw_marks(image, marks, color=(0, 255, 0)):
    """utils/plots.py:152-166: a filled radius-2 disc at every landmark, in place (the reference mutates the frame
    too); points whose disc leaves the image are clipped."""
    h, w = image.shape[:2]
    for mark in np.asarray(marks).reshape(-1, 2):
        x, y = int(mark[0]), int(mark[1])
        for dx, dy in _DISC2:
            xx, yy = x + dx, y + dy
            if 0 <= xx < w and 0 <= yy < h:
                image[yy, xx] = color
    return image
