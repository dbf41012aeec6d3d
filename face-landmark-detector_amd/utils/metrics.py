"""Heatmap -> (x, y) decode on the device: mirror of reference utils/metrics.py:46-109.

Same names and argument meaning.  The reference's `transfer_xy_coord` passes
`(n_points, thresh)` into `get_average_xy`'s `(height, width)` slots (:98), so as shipped
every call behaves as n_points=4, thresh=0.  This module honours the documented arguments;
`as_shipped=True` (or the module switch below) reproduces the shipped behaviour exactly.
Device limits: 1 <= n_points <= 64 in top-n mode, <= 96 landmarks per map.
"""
from __future__ import annotations

import numpy as np

from .. import _lib

REFERENCE_POSITIONAL_SLIP = False


def decode_device(hm, n_points=4, thresh=0.0, out=None):
    """hm: CUDA float32 [N,H,W,L] contiguous -> CUDA float64 [N,L,2] (x,y)."""
    import torch
    lib = _lib.load()
    if hm.dim() != 4 or hm.dtype != torch.float32 or not hm.is_cuda or not hm.is_contiguous():
        raise ValueError("decode_device needs a contiguous CUDA float32 [N,H,W,L] tensor")
    n, h, w, l = [int(v) for v in hm.shape]
    if n == 0:
        return torch.empty((0, l, 2), dtype=torch.float64, device=hm.device)
    mode, npts = (_lib.DECODE_ALL, 0) if n_points < 1 else (_lib.DECODE_TOPN, int(n_points))
    nbytes = lib.flm_decode_workspace_bytes(n, h, w, l, mode, npts)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=hm.device)
    if out is None:
        out = torch.empty((n, l, 2), dtype=torch.float64, device=hm.device)
    _lib.check(lib.flm_decode(_lib.stream_ptr(), _lib.ptr(hm), n, h, w, l, mode, npts, float(thresh),
                              _lib.ptr(out), _lib.ptr(ws), ws.numel()), "flm_decode")
    return out


def _to_device(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(_lib.require_gpu())


def get_average_xy(hmi, height=96, width=96, n_points=4, thresh=0):
    """utils/metrics.py:46-80.  `height`/`width` must equal the map's dims (the reference builds
    its index grids from them, :61-63); returns [x, y]."""
    hmi = np.asarray(hmi)
    if hmi.ndim != 2:
        raise ValueError("hmi must be 2-D")
    if n_points < 1 and tuple(hmi.shape) != (height, width):
        raise ValueError("height/width must equal hmi.shape in all-pixel mode (utils/metrics.py:61-63)")
    xy = decode_device(_to_device(hmi[None, :, :, None]), n_points, thresh).cpu().numpy()[0, 0]
    return [xy[0], xy[1]]


def transfer_xy_coord(hm, n_points=64, thresh=0.2, as_shipped=None):
    """utils/metrics.py:83-99: [H,W,L] -> list of 2L floats (x0,y0,x1,y1,...)."""
    hm = np.asarray(hm)
    assert len(hm.shape) == 3
    return list(transfer_target(hm[None], thresh, n_points, as_shipped)[0])


def transfer_target(y_pred, thresh=0, n_points=64, as_shipped=None):
    """utils/metrics.py:102-109: [N,H,W,L] -> float64 [N, 2L]."""
    if as_shipped is None:
        as_shipped = REFERENCE_POSITIONAL_SLIP
    if as_shipped:
        n_points, thresh = 4, 0
    import torch
    if isinstance(y_pred, torch.Tensor):
        hm = y_pred if y_pred.is_cuda else y_pred.to(_lib.require_gpu())
        hm = hm.contiguous().float()
    else:
        hm = _to_device(np.asarray(y_pred))
    out = decode_device(hm, n_points, thresh)
    return out.reshape(out.shape[0], 2 * out.shape[1]).cpu().numpy()


def get_RMSE(y_pred_xy, y_train_xy, pick_not_NA):
    """utils/metrics.py:112-115 (host numpy, unchanged semantics)."""
    res = y_pred_xy[pick_not_NA] - y_train_xy[pick_not_NA]
    return np.sqrt(np.mean(res ** 2))
