"""Heatmap -> (x, y) decode on the device: mirror of reference utils/metrics.py:46-109.

Same names and argument meaning.  The reference's `transfer_xy_coord` passes
`(n_points, thresh)` into `get_average_xy`'s `(height, width)` slots (:98), so as shipped
EVERY call decodes with n_points=4, thresh=0, whatever the caller passed.  Drop-in rule here:
  * a call that relies on the defaults -- `transfer_target(y)`, `transfer_xy_coord(hm)` -- returns
    what the reference returns (top-4, thresh 0);
  * a call that passes `n_points` / `thresh` explicitly gets what it asked for (the reference would
    silently ignore the arguments) and a one-time warning says so;
  * `as_shipped=True` forces the reference's behaviour, `as_shipped=False` the documented one.
Device limits: 1 <= n_points <= 128 in top-n mode (the reference's own sweep reaches 81, utils/metrics.py:130-133),
<= 96 landmarks per map.
"""
from __future__ import annotations

import warnings

import numpy as np

from .. import _lib

_UNSET = object()
_warned = False


def _resolve(n_points, thresh, as_shipped, dflt_n, dflt_t):
    """(n_points, thresh) a decode call runs with; see the module docstring."""
    global _warned
    explicit = n_points is not _UNSET or thresh is not _UNSET
    if as_shipped is None:
        as_shipped = not explicit
        if explicit and not _warned:
            _warned = True
            warnings.warn("flm_amd.utils.metrics: n_points / thresh are honoured here; the reference as shipped ignores "
                          "them and always decodes top-4 with thresh 0 (utils/metrics.py:98) -- pass as_shipped=True "
                          "for that behaviour", stacklevel=3)
    if as_shipped:
        return 4, 0
    return (dflt_n if n_points is _UNSET else n_points), (dflt_t if thresh is _UNSET else thresh)


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


def transfer_xy_coord(hm, n_points=_UNSET, thresh=_UNSET, as_shipped=None):
    """utils/metrics.py:83-99 (documented defaults n_points=64, thresh=0.2): [H,W,L] -> list of 2L floats."""
    hm = np.asarray(hm)
    assert len(hm.shape) == 3
    n_points, thresh = _resolve(n_points, thresh, as_shipped, 64, 0.2)
    return list(transfer_target(hm[None], thresh, n_points, as_shipped=False)[0])


def transfer_target(y_pred, thresh=_UNSET, n_points=_UNSET, as_shipped=None):
    """utils/metrics.py:102-109 (documented defaults thresh=0, n_points=64): [N,H,W,L] -> float64 [N, 2L]."""
    n_points, thresh = _resolve(n_points, thresh, as_shipped, 64, 0)
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
