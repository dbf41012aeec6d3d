"""CPU restatement of the alignment path (similarity estimate + bilinear warp) and of the crop
front-end.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against the reference: it contains no alignment code (README.md:1 states the
intent only).  The spec is build-defined and written operation by
operation in the kernel comments of csrc/flm_misc.hip; this file restates it in numpy.
`tests/test_oracle_warp.py` cross-checks the geometry against scikit-image 0.18.3
(`SimilarityTransform.estimate` + `warp(order=1, mode="edge")`, the call shape of the
reference's only affine warp, data/generator.py:192-200) through fixtures generated with the
second interpreter in the build container.

fp32 fused multiply-add is restated EXACTLY: the product of two float32 is exact in float64; the sum with the
addend is taken with its rounding error (TwoSum), and where the float64 sum sits exactly on a float32 rounding
boundary the error term decides the direction -- so `fma()` below returns the correctly rounded float32 of a*b+c, as
the hardware instruction does.  The crop/resize is integer fixed point on both sides (bit-exact by construction).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def fma(a, b, c):
    """Correctly rounded float32 fused multiply-add of float32 operands (see the module docstring)."""
    a, b, c = (np.asarray(v, np.float32).astype(np.float64) for v in (a, b, c))
    p = a * b                                   # exact: 24 + 24 significant bits
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)             # TwoSum: p + c == s + err exactly
    r = s.astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        hi = np.nextafter(r, np.float32(np.inf))
        lo = np.nextafter(r, np.float32(-np.inf))
        r64 = r.astype(np.float64)
        tie_hi = (s == (r64 + hi.astype(np.float64)) / 2) & (err > 0)   # exact value lies above the midpoint
        tie_lo = (s == (r64 + lo.astype(np.float64)) / 2) & (err < 0)
    return np.where(tie_hi, hi, np.where(tie_lo, lo, r)).astype(np.float32)


def similarity_ref(lm: np.ndarray, tmpl: np.ndarray) -> np.ndarray:
    """Least-squares similarity (rotation+uniform scale+translation, no reflection) mapping
    landmarks [N,K,2] onto template [K,2]; float64 sequential sums; returns float32 [N,2,3].
    Landmarks with a negative coordinate (decode rejects, utils/metrics.py:78-79) are left out."""
    n, k, _ = lm.shape
    out = np.zeros((n, 2, 3), np.float32)
    for f in range(n):
        p = lm[f].astype(np.float64)
        ok = [i for i in range(k) if not (p[i, 0] < 0.0 or p[i, 1] < 0.0)]
        a, b, tx, ty = 1.0, 0.0, 0.0, 0.0
        if len(ok) >= 2:
            mpx = mpy = mqx = mqy = 0.0
            for i in ok:
                mpx += p[i, 0]; mpy += p[i, 1]; mqx += float(tmpl[i, 0]); mqy += float(tmpl[i, 1])
            cnt = len(ok)
            mpx /= cnt; mpy /= cnt; mqx /= cnt; mqy /= cnt
            sa = sb = var = 0.0
            for i in ok:
                px, py = p[i, 0] - mpx, p[i, 1] - mpy
                qx, qy = float(tmpl[i, 0]) - mqx, float(tmpl[i, 1]) - mqy
                sa += px * qx + py * qy
                sb += px * qy - py * qx
                var += px * px + py * py
            if var > 0.0:
                a, b = sa / var, sb / var
                tx = mqx - (a * mpx - b * mpy)
                ty = mqy - (b * mpx + a * mpy)
        out[f] = [[a, -b, tx], [b, a, ty]]
    return out


def _bilinear(src_f32, xs, ys):
    hs, ws = src_f32.shape[:2]
    xs = np.minimum(np.maximum(xs, f32(0)), f32(ws - 1))
    ys = np.minimum(np.maximum(ys, f32(0)), f32(hs - 1))
    xf, yf = np.floor(xs), np.floor(ys)
    fx, fy = (xs - xf).astype(np.float32), (ys - yf).astype(np.float32)
    x0, y0 = xf.astype(np.int64), yf.astype(np.int64)
    x1, y1 = np.minimum(x0 + 1, ws - 1), np.minimum(y0 + 1, hs - 1)
    p00, p01 = src_f32[y0, x0], src_f32[y0, x1]
    p10, p11 = src_f32[y1, x0], src_f32[y1, x1]
    top = fma(fx[..., None], p01 - p00, p00)
    bot = fma(fx[..., None], p11 - p10, p10)
    return fma(fy[..., None], bot - top, top)


def warp_affine_ref(src: np.ndarray, m: np.ndarray, hd: int, wd: int) -> np.ndarray:
    """src [N,Hs,Ws,3] uint8 or float32, m float32 [N,2,3] (source -> aligned coords);
    returns float32 [N,hd,wd,3].  Inverse map + bilinear + edge clamp (csrc/flm_misc.hip)."""
    n = src.shape[0]
    out = np.zeros((n, hd, wd, 3), np.float32)
    yd, xd = np.mgrid[0:hd, 0:wd]
    xd, yd = xd.astype(np.float32), yd.astype(np.float32)
    for f in range(n):
        m00, m01, m02, m10, m11, m12 = [f32(v) for v in m[f].reshape(-1)]
        det = fma(m00, m11, -(m01 * m10))
        idet = f32(1.0) / det
        i00, i01, i10, i11 = m11 * idet, -m01 * idet, -m10 * idet, m00 * idet
        i02 = -fma(i00, m02, i01 * m12)
        i12 = -fma(i10, m02, i11 * m12)
        xs = fma(i00, xd, fma(i01, yd, i02))
        ys = fma(i10, xd, fma(i11, yd, i12))
        out[f] = _bilinear(src[f].astype(np.float32), xs, ys)
    return out


def _resize_coef(n_dst: int, n_src: int):
    """Per destination index: (s0, s1, w0, w1) of OpenCV's 8-bit INTER_LINEAR (imgproc/resize.cpp, generic path):
    scale = 1/(n_dst/n_src) in double; f = float((d + 0.5)*scale - 0.5); s = floor(f); f -= s; clamped at both ends;
    11-bit weights w1 = rint(f*2048), w0 = rint((1-f)*2048) (cvRound: ties to even)."""
    scale = np.float64(1.0) / (np.float64(n_dst) / np.float64(n_src))
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    low, high = s < 0, s >= n_src - 1
    s = np.where(low, 0, np.where(high, n_src - 1, s))
    f = np.where(low | high, f32(0), f).astype(np.float32)
    w0 = np.rint((f32(1) - f) * f32(2048)).astype(np.int64)
    w1 = np.rint(f * f32(2048)).astype(np.int64)
    return s, np.minimum(s + 1, n_src - 1), w0, w1


def _resize_coef_y(n_dst: int, n_src: int):
    """The same along y, where OpenCV clamps only the row indices (`clip(sy + k, 0, h)`) and keeps the split weights of
    the unclamped position: on the border rows of an upscale both taps read the border row, weighted b0 and b1."""
    scale = np.float64(1.0) / (np.float64(n_dst) / np.float64(n_src))
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    w0 = np.rint((f32(1) - f) * f32(2048)).astype(np.int64)
    w1 = np.rint(f * f32(2048)).astype(np.int64)
    return np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1), w0, w1


def resize_u8_ref(src: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """`cv2.resize(src, (ow, oh))` for uint8 [h,w,c] with the default INTER_LINEAR, as OpenCV's generic code computes
    it (data/generator.py:53, prediction.py:82).  Integer arithmetic throughout:
      horizontal  h = S[x0]*a0 + S[x1]*a1                      (int32, weights sum to 2048)
      vertical    out = (((b0*(h0 >> 4)) >> 16) + ((b1*(h1 >> 4)) >> 16) + 2) >> 2
      exact 2x downscale in both axes -> INTER_AREA: (S00 + S01 + S10 + S11 + 2) >> 2."""
    ch, cw = src.shape[:2]
    s = src.astype(np.int64)
    if cw == 2 * ow and ch == 2 * oh:
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = _resize_coef(ow, cw)
    y0, y1, b0, b1 = _resize_coef_y(oh, ch)
    hrow = s[:, x0] * a0[None, :, None] + s[:, x1] * a1[None, :, None]            # [ch, ow, c]
    h0, h1 = hrow[y0] >> 4, hrow[y1] >> 4
    out = (((b0[:, None, None] * h0) >> 16) + ((b1[:, None, None] * h1) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def crop_resize_ref(frame: np.ndarray, boxes: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """frame [H,W,3] uint8, boxes int [K,4] (x0,y0,x1,y1) -> uint8 [K,oh,ow,3]: `img[y0:y1, x0:x1]` then
    `cv2.resize(..., (ow, oh))` (prediction.py:80-82); the box is clipped to the frame (csrc/flm_misc.hip), an empty
    intersection gives zeros."""
    fh, fw = frame.shape[:2]
    out = np.zeros((boxes.shape[0], oh, ow, 3), np.uint8)
    for i in range(boxes.shape[0]):
        bx0, by0, bx1, by1 = [int(v) for v in boxes[i]]
        cx0, cx1 = min(max(bx0, 0), fw), min(max(bx1, 0), fw)
        cy0, cy1 = min(max(by0, 0), fh), min(max(by1, 0), fh)
        if cx1 > cx0 and cy1 > cy0:
            out[i] = resize_u8_ref(frame[cy0:cy1, cx0:cx1], oh, ow)
    return out


# ---- detect_marks box maths (reference prediction.py:36-94), integer exact -------------------------
def square_box_ref(face):
    """offset_y = int(abs((y1-y0)*0.1)); move down; square by symmetric integer expansion."""
    offset_y = int(abs((face[3] - face[1]) * 0.1))                      # :76
    left_x, top_y, right_x, bottom_y = face[0], face[1] + offset_y, face[2], face[3] + offset_y  # :67-74,77
    box_width, box_height = right_x - left_x, bottom_y - top_y          # :40-41
    diff = box_height - box_width                                       # :44
    delta = int(abs(diff) / 2)                                          # :45
    if diff == 0:
        return [left_x, top_y, right_x, bottom_y]
    elif diff > 0:                                                      # :50-54
        left_x -= delta
        right_x += delta
        if diff % 2 == 1:
            right_x += 1
    else:                                                               # :56-60
        top_y -= delta
        bottom_y += delta
        if diff % 2 == 1:
            bottom_y += 1
    assert (right_x - left_x) == (bottom_y - top_y)
    return [left_x, top_y, right_x, bottom_y]


def backproject_marks_ref(marks01: np.ndarray, facebox) -> np.ndarray:
    """prediction.py:91-94: marks in [0,1] -> image coords, truncated to unsigned ints."""
    marks = np.array(marks01, dtype=np.float32).reshape(-1, 2).copy()
    marks *= (facebox[2] - facebox[0])
    marks[:, 0] += facebox[0]
    marks[:, 1] += facebox[1]
    return marks.astype(np.uint)
