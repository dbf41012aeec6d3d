"""CPU restatement of the reference's heatmap -> (x, y) decode.  TEST INFRASTRUCTURE ONLY.

PINNED: `tests/golden/make_decode_golden.py` imports the reference's
`keypoints_detector/utils/metrics.py` in the build container, runs it on seeded
heatmaps and freezes inputs + outputs in `tests/golden/decode_golden.npz`;
`tests/test_oracle_decode.py` checks this file against that fixture (and, when
`/root/reference` is present, against the live import).

Follows utils/metrics.py:46-109.  numpy dtype behaviour that the restatement
keeps (NumPy >= 2 promotion rules, the version installed here):

* top-n branch (:66-77): `hsum` starts as Python int 0 and accumulates float32
  scalars -> a sequential FLOAT32 sum in ascending-value order; `i0`/`i1`
  accumulate `np.int64 * np.float32` products -> float64.
* all-pixel branch (:58-64): `np.sum(hmi)` of a float32 map is numpy's float32
  pairwise sum; the index-weighted sums are int64*float32 -> float64.
* reject test (:78-79): `hsum / n_points <= thresh` evaluated in hsum's dtype.

Tie rule (the reference's `argsort` is an unstable sort, so ties at the n-th
place are unspecified there): this build orders pixels by (value, flat index)
and keeps the n largest, i.e. `argsort(kind="stable")[-n:]`.
"""
from __future__ import annotations

import numpy as np


def get_average_xy_ref(hmi: np.ndarray, n_points: int = 4, thresh=0):
    """utils/metrics.py:46-80 with `height, width` taken from the map itself
    (the reference builds its index grids from those two arguments, :61-63, so
    they must equal the map's dims for the call to be meaningful)."""
    height, width = hmi.shape
    if n_points < 1:
        hsum, n_points = np.sum(hmi), hmi.size                               # :60
        ind = np.arange(width, dtype=np.int64)[None, :].repeat(height, 0)    # :61
        with np.errstate(invalid="ignore", divide="ignore"):
            i1 = np.sum(ind * hmi) / hsum                                    # :62
            ind = np.arange(height, dtype=np.int64)[:, None].repeat(width, 1)  # :63
            i0 = np.sum(ind * hmi) / hsum                                    # :64
    else:
        ind = hmi.argsort(axis=None, kind="stable")[-n_points:]              # :66 (+ tie rule)
        top0, top1 = np.unravel_index(ind, hmi.shape)                        # :67
        i0, i1, hsum = 0, 0, 0                                               # :69
        for a, b in zip(top0, top1):                                         # :70-74
            h = hmi[a, b]
            hsum += h
            i0 += a * h
            i1 += b * h
        with np.errstate(invalid="ignore", divide="ignore"):
            i0 /= hsum                                                       # :76
            i1 /= hsum                                                       # :77
    if hsum / n_points <= thresh:                                            # :78-79
        i0, i1 = -1, -1
    return [i1, i0]                                                          # :80


def transfer_xy_coord_ref(hm: np.ndarray, n_points: int = 64, thresh=0.2, as_shipped: bool = False):
    """utils/metrics.py:83-99.  `as_shipped=True` reproduces the positional-argument
    slip at :98 (`get_average_xy(hmi, n_points, thresh)` binds height/width), which
    makes every call behave as n_points=4, thresh=0."""
    assert hm.ndim == 3
    est = []
    for i in range(hm.shape[-1]):
        if as_shipped:
            est.extend(get_average_xy_ref(hm[:, :, i], 4, 0))
        else:
            est.extend(get_average_xy_ref(hm[:, :, i], n_points, thresh))
    return est


def transfer_target_ref(y_pred: np.ndarray, thresh=0, n_points: int = 64, as_shipped: bool = False) -> np.ndarray:
    """utils/metrics.py:102-109: [N,H,W,L] -> float64 [N, 2L] (x0,y0,x1,y1,...)."""
    return np.array([transfer_xy_coord_ref(y_pred[i], n_points, thresh, as_shipped)
                     for i in range(y_pred.shape[0])], dtype=np.float64)


def topn_gap_rel(maps: np.ndarray, n: int) -> np.ndarray:
    """maps [HW, L] -> [L]: relative gap (v_n - v_{n+1}) / v_n between the n-th and (n+1)-th largest values of every
    channel.  A top-n centroid SELECTS pixels: where two evaluations of the network disagree by more than this gap at
    the n-th place they may select different pixels, and the coordinates then differ by the selection, not by rounding
    (tests/test_gpu_baseline_configs.py calls such (face, class) pairs undetermined).  Checker helper, not a
    restatement of reference code."""
    hw = maps.shape[0]
    part = np.partition(maps, hw - n - 1, axis=0)[hw - n - 1:]      # the n + 1 largest, the smallest of them first
    vn1 = part[0]
    vn = np.partition(part[1:], 0, axis=0)[0]
    with np.errstate(invalid="ignore", divide="ignore"):
        return (vn - vn1) / vn
