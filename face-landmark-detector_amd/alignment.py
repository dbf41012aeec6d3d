"""Face alignment: similarity transform from predicted landmarks + bilinear warp, on the device.

The reference states the intent ("predict landmark and align face for Face Match",
README.md:1) but contains no alignment code; this is build-defined (SURVEY.md section 8 row
A8).  The warp follows the shape of the reference's only affine warp
(`skimage.transform.warp(im, tform, mode="edge")`, data/generator.py:192-200): inverse map,
bilinear, edge clamp -- but keeps pixel units (no [0,1] rescale) and float32.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def canonical_template(n_landmarks: int, out_h: int, out_w: int) -> np.ndarray:
    """Deterministic canonical landmark layout, float64 [K,2] (x,y) in output pixels.

    68 landmarks: the iBUG-68 ordering laid out procedurally (jaw 0-16, brows 17-26, nose
    27-35, eyes 36-47, mouth 48-67) inside the unit square; other counts: a centred ellipse.
    No trained model exists for the reference (README.md:4-7), so the template only has to be
    a fixed, well-conditioned target for the similarity fit.
    """
    k = int(n_landmarks)
    pts = np.zeros((k, 2), np.float64)
    if k == 68:
        t = np.linspace(np.pi * 1.05, np.pi * 1.95, 17)       # jaw: lower arc, left to right
        pts[0:17] = np.stack([0.5 + 0.42 * np.cos(t), 0.42 - 0.48 * np.sin(t)], 1)
        pts[17:22] = np.stack([np.linspace(0.18, 0.42, 5), 0.30 - 0.03 * np.sin(np.linspace(0, np.pi, 5))], 1)
        pts[22:27] = np.stack([np.linspace(0.58, 0.82, 5), 0.30 - 0.03 * np.sin(np.linspace(0, np.pi, 5))], 1)
        pts[27:31] = np.stack([np.full(4, 0.5), np.linspace(0.38, 0.56, 4)], 1)
        pts[31:36] = np.stack([np.linspace(0.42, 0.58, 5), 0.62 + 0.015 * np.sin(np.linspace(0, np.pi, 5))], 1)
        for base, cx in ((36, 0.31), (42, 0.69)):
            a = np.linspace(np.pi, -np.pi, 7)[:6]
            pts[base:base + 6] = np.stack([cx + 0.07 * np.cos(a), 0.40 - 0.03 * np.sin(a)], 1)
        a = np.linspace(np.pi, -np.pi, 13)[:12]
        pts[48:60] = np.stack([0.5 + 0.13 * np.cos(a), 0.76 - 0.06 * np.sin(a)], 1)
        a = np.linspace(np.pi, -np.pi, 9)[:8]
        pts[60:68] = np.stack([0.5 + 0.08 * np.cos(a), 0.76 - 0.025 * np.sin(a)], 1)
    else:
        a = np.linspace(0, 2 * np.pi, k, endpoint=False)
        pts = np.stack([0.5 + 0.35 * np.cos(a), 0.5 + 0.40 * np.sin(a)], 1)
    return pts * np.array([out_w - 1, out_h - 1], np.float64)


def similarity_device(landmarks, template, landmark_scale=(1.0, 1.0)):
    """landmarks: CUDA float64 [N,K,2]; template: CUDA float64 [K,2] -> CUDA float32 [N,2,3].
    `landmark_scale` (sx, sy) takes the landmarks to the template's pixel units inside the kernel (float64
    products; the decode's reject marker (-1,-1) stays negative, so the fit skips those points)."""
    import torch
    lib = _lib.load()
    n, k, _ = landmarks.shape
    if landmarks.dtype != torch.float64 or template.dtype != torch.float64:
        raise ValueError("landmarks and template must be float64")
    if tuple(template.shape) != (k, 2):
        raise ValueError("template must be [K,2]")
    m = torch.empty((n, 2, 3), dtype=torch.float32, device=landmarks.device)
    _lib.check(lib.flm_similarity_from_landmarks_scaled(_lib.stream_ptr(), _lib.ptr(landmarks.contiguous()),
                                                        _lib.ptr(template.contiguous()), n, k,
                                                        float(landmark_scale[0]), float(landmark_scale[1]),
                                                        _lib.ptr(m)),
               "flm_similarity_from_landmarks_scaled")
    return m


def warp_device(src, m, out_h, out_w, out=None):
    """src: CUDA uint8/float32 [N,Hs,Ws,3]; m: CUDA float32 [N,2,3] (source -> aligned)."""
    import torch
    lib = _lib.load()
    if src.dim() != 4 or src.shape[3] != 3 or src.dtype not in (torch.uint8, torch.float32):
        raise ValueError("src must be uint8/float32 [N,H,W,3]")
    n, hs, ws, _ = [int(v) for v in src.shape]
    if out is None:
        out = torch.empty((n, out_h, out_w, 3), dtype=torch.float32, device=src.device)
    _lib.check(lib.flm_warp_affine(_lib.stream_ptr(), _lib.ptr(src.contiguous()), int(src.dtype == torch.uint8),
                                   n, hs, ws, _lib.ptr(m.contiguous()), _lib.ptr(out), out_h, out_w),
               "flm_warp_affine")
    return out


def align_device(crops, landmarks_in, template, out_h, out_w, landmark_scale=(1.0, 1.0)):
    """crops [N,H,W,3] + landmarks (crop pixel units after `landmark_scale`) -> aligned crops, M."""
    m = similarity_device(landmarks_in, template, landmark_scale)
    return warp_device(crops, m, out_h, out_w), m
