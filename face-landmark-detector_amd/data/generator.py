"""`get_image_array` of the predict path (reference data/generator.py:29-69) on the device.

Same signature, argument meaning and errors.  cv2 is not a dependency: files are decoded
with PIL into BGR (what cv2.imread returns), and an image whose size differs from
(width, height) is resampled on the GPU by `flm_crop_resize`: OpenCV's 8-bit INTER_LINEAR
restated in integer fixed point (11-bit weights, the exact 2x downscale as INTER_AREA), bit-equal
to `oracle/warp_ref.resize_u8_ref`; parity with the cv2 binary itself is unpinned (cv2 is not
installable here).  For crops already at model size -- the hot path -- the resize is the identity,
as in cv2.
"""
from __future__ import annotations

import os

import numpy as np
import six

from .. import _lib


class DataLoaderError(Exception):
    pass


_NORMS = {"sub_mean": _lib.NORM_SUB_MEAN, "sub_and_divide": _lib.NORM_SUB_AND_DIVIDE, "divide": _lib.NORM_DIVIDE}


def imread_bgr(path, read_image_type=1):
    """cv2.imread(path, 1) stand-in: uint8 HxWx3 in BGR order (or HxW for type 0)."""
    from PIL import Image
    im = Image.open(path)
    if read_image_type == 0:
        return np.asarray(im.convert("L"))
    return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


def resize_u8_device(img_dev, height, width):
    """[H,W,3] uint8 CUDA tensor -> [height,width,3] via flm_crop_resize on the full frame."""
    import torch
    lib = _lib.load()
    h, w = int(img_dev.shape[0]), int(img_dev.shape[1])
    if (h, w) == (height, width):
        return img_dev
    boxes = torch.tensor([[0, 0, w, h]], dtype=torch.int32, device=img_dev.device)
    out = torch.empty((1, height, width, 3), dtype=torch.uint8, device=img_dev.device)
    _lib.check(lib.flm_crop_resize(_lib.stream_ptr(), _lib.ptr(img_dev), h, w, _lib.ptr(boxes), 1,
                                   _lib.ptr(out), height, width), "flm_crop_resize")
    return out[0]


def get_image_array(image, width, height, imgNorm="sub_mean", ordering="channels_first", read_image_type=1):
    """Load image array from input (reference data/generator.py:29-69)."""
    import torch
    if isinstance(image, np.ndarray):
        img = image
    elif isinstance(image, six.string_types):
        if not os.path.isfile(image):
            raise DataLoaderError("get_image_array: path {0} doesn't exist".format(image))
        img = imread_bgr(image, read_image_type)
    else:
        raise DataLoaderError("get_image_array: Can't process input type {0}".format(str(type(image))))
    if imgNorm not in _NORMS:
        return img  # the reference falls through every branch and returns the raw image
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
        raise DataLoaderError("get_image_array: the device path takes uint8 HxWx3 images")
    lib = _lib.load()
    dev = _lib.require_gpu()
    d = resize_u8_device(torch.from_numpy(np.ascontiguousarray(img)).to(dev), height, width)
    out = torch.empty((height, width, 3), dtype=torch.float32, device=dev)
    _lib.check(lib.flm_preprocess(_lib.stream_ptr(), _lib.ptr(d.contiguous()), 1, height, width, _NORMS[imgNorm],
                                  _lib.ptr(out)), "flm_preprocess")
    res = out.cpu().numpy()
    if ordering == "channels_first":
        res = np.rollaxis(res, 2, 0)
    return res
