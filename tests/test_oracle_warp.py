"""Cross-check oracle/warp_ref.py against scikit-image 0.18.3 outputs (tests/golden/warp_golden.npz,
made by tests/golden/make_warp_golden.py) and pin the detect_marks box maths.

The reference has no alignment code (parity unpinned); skimage is the library its only affine
warp calls (data/generator.py:192-200).  skimage computes in float64, the spec in float32:
the comparison tolerance covers the fp32 coordinate rounding (1e-5 px times the image gradient).
"""
import os

import numpy as np
import pytest

from oracle import warp_ref


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "warp_golden.npz"))


def test_similarity_matches_skimage_umeyama(gold):
    m = warp_ref.similarity_ref(gold["src_pts"], gold["dst_pts"][0])  # shape check only
    assert m.shape == (4, 2, 3)
    for i in range(4):
        mi = warp_ref.similarity_ref(gold["src_pts"][i:i + 1], gold["dst_pts"][i])[0]
        np.testing.assert_allclose(mi, gold["mats"][i][:2], rtol=2e-6, atol=2e-5)


def test_warp_matches_skimage(gold):
    n, hd, wd = 4, 40, 44
    m = gold["mats"][:, :2, :].astype(np.float32)
    got = warp_ref.warp_affine_ref(gold["imgs"], m, hd, wd)
    assert got.dtype == np.float32 and got.shape == (n, hd, wd, 3)
    assert np.abs(got - gold["warped"]).max() < 0.05


def test_fma_is_correctly_rounded():
    """oracle fma == the hardware's: a crafted double-rounding case (float64 sum lands exactly on a float32 rounding
    boundary with the lost bits below it) and random operands against exact rational arithmetic."""
    from fractions import Fraction
    a, b, c = np.float32(2 ** 15 + 2 ** -8), np.float32(2 ** 15 - 2 ** -8), np.float32(2 ** 54 + 2 ** 31)
    naive = np.float32(np.float64(a) * np.float64(b) + np.float64(c))
    assert float(naive) == 2.0 ** 54 + 2.0 ** 32                       # what rounding twice gives
    assert float(warp_ref.fma(a, b, c)) == 2.0 ** 54 + 2.0 ** 31       # a*b + c = 2^54 + 2^31 + 2^30 - 2^-16
    rng = np.random.default_rng(3)
    a = rng.standard_normal(400).astype(np.float32)
    b = rng.standard_normal(400).astype(np.float32)
    c = (-(a.astype(np.float64) * b) + rng.standard_normal(400) * 1e-4).astype(np.float32)   # heavy cancellation
    r = warp_ref.fma(a, b, c)
    for i in range(400):
        ex = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
        near = np.float32(float(ex))
        cands = (np.nextafter(near, np.float32(-np.inf)), near, np.nextafter(near, np.float32(np.inf)))
        best = min(cands, key=lambda v: (abs(Fraction(float(v)) - ex), int(np.float32(v).view(np.uint32)) & 1))
        assert r[i] == best


def test_resize_u8_fixed_point_properties():
    """oracle/warp_ref.resize_u8_ref (OpenCV's 8-bit INTER_LINEAR restated; cv2 is absent, so parity with the library
    itself is unpinned): identity at equal size, constants stay constant, the 2x area path, hand-worked weights."""
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (30, 44, 3), dtype=np.uint8)
    assert np.array_equal(warp_ref.resize_u8_ref(img, 30, 44), img)
    flat = np.full((17, 23, 3), 201, np.uint8)
    assert (warp_ref.resize_u8_ref(flat, 40, 9) == 201).all()
    blk = img.astype(np.int32)
    area = (blk[0::2, 0::2] + blk[0::2, 1::2] + blk[1::2, 0::2] + blk[1::2, 1::2] + 2) >> 2
    assert np.array_equal(warp_ref.resize_u8_ref(img, 15, 22), area.astype(np.uint8))
    # 2 -> 4 upscale of a row [0, 200]: f = (d+.5)/2 - .5 = -.25, .25, .75, 1.25 -> weights 2048|0, 1536|512, 512|1536, 2048|0
    row = np.array([[[0], [200]]], np.uint8)
    out = warp_ref.resize_u8_ref(row, 1, 4)[0, :, 0]
    assert out.tolist() == [0, 50, 150, 200]
    s0, s1, w0, w1 = warp_ref._resize_coef(4, 2)
    assert s0.tolist() == [0, 0, 0, 1] and w1.tolist() == [0, 512, 1536, 0] and (w0 + w1 == 2048).all()


def test_resize_u8_border_rows_keep_split_weights():
    """Along y OpenCV's generic 8-bit resize clamps only the row indices (`clip(sy + k, 0, h)`, imgproc/resize.cpp) and
    keeps the split weights of the unclamped position, while along x it folds them (fx = 0 at the border).  On the
    first and last output rows of an upscale both taps therefore read the border row with weights b0 | b1, and
    floor(b0*v >> 16) + floor(b1*v >> 16) differs from (2048*v) >> 16 by one LSB for some horizontal sums v.  The border
    rows are written out by hand here, independently of the vectorised restatement (the x weights are taken from it)."""
    rng = np.random.default_rng(12)
    for (h, oh) in ((96, 256), (150, 256), (200, 256)):
        img = rng.integers(0, 256, (h, h, 1), dtype=np.uint8)
        got = warp_ref.resize_u8_ref(img, oh, oh)
        x0, x1, a0, a1 = warp_ref._resize_coef(oh, h)
        scale = 1.0 / (oh / h)
        differs_from_folded = 0
        for dy in (0, 1, oh - 2, oh - 1):
            f = np.float32((dy + 0.5) * scale - 0.5)
            sy = int(np.floor(f))
            fy = np.float32(f - np.float32(sy))
            b1 = int(np.rint(fy * np.float32(2048)))
            b0 = int(np.rint((np.float32(1) - fy) * np.float32(2048)))
            r0, r1 = min(max(sy, 0), h - 1), min(max(sy + 1, 0), h - 1)
            for x in range(oh):
                h0 = int(img[r0, x0[x], 0]) * int(a0[x]) + int(img[r0, x1[x], 0]) * int(a1[x])
                h1 = int(img[r1, x0[x], 0]) * int(a0[x]) + int(img[r1, x1[x], 0]) * int(a1[x])
                exp = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2
                assert int(got[dy, x, 0]) == exp, (h, oh, dy, x)
                if r0 == r1:
                    differs_from_folded += exp != ((((2048 * (h0 >> 4)) >> 16) + 2) >> 2)
        assert differs_from_folded > 0          # the case the folded rule got wrong is really exercised
    s0, s1, w0, w1 = warp_ref._resize_coef_y(4, 2)
    assert s0.tolist() == [0, 0, 0, 1] and s1.tolist() == [0, 1, 1, 1] and w1.tolist() == [1536, 512, 1536, 512]


def test_similarity_skips_rejected_landmarks():
    rng = np.random.default_rng(0)
    p = rng.uniform(10, 200, (1, 10, 2))
    tm = p[0] * 0.5 + 3.0
    p2 = p.copy()
    p2[0, 3] = [-1, -1]  # decode reject marker
    a = warp_ref.similarity_ref(p, tm)[0]
    b = warp_ref.similarity_ref(p2, tm)[0]
    np.testing.assert_allclose(a, [[0.5, 0, 3], [0, 0.5, 3]], atol=1e-5)
    np.testing.assert_allclose(b, [[0.5, 0, 3], [0, 0.5, 3]], atol=1e-5)
    p3 = np.full((1, 10, 2), -1.0)
    np.testing.assert_array_equal(warp_ref.similarity_ref(p3, tm)[0], [[1, 0, 0], [0, 1, 0]])


def test_identity_warp_is_exact_copy():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (2, 9, 11, 3), dtype=np.uint8)
    m = np.tile(np.array([[1, 0, 0], [0, 1, 0]], np.float32), (2, 1, 1))
    np.testing.assert_array_equal(warp_ref.warp_affine_ref(img, m, 9, 11), img.astype(np.float32))


def test_square_box_follows_reference_lines():
    # prediction.py:36-78 worked by hand
    assert warp_ref.square_box_ref([10, 20, 110, 120]) == [10, 30, 110, 130]          # square, offset 10
    assert warp_ref.square_box_ref([10, 20, 60, 120]) == [-15, 30, 85, 130]           # tall: widen by 25 each side
    assert warp_ref.square_box_ref([10, 20, 61, 120]) == [-14, 30, 86, 130]           # odd diff 49: 24 each side, +1 on the right
    assert warp_ref.square_box_ref([0, 0, 100, 41]) == [0, -25, 100, 75]              # wide, odd diff: +1 bottom
    b = warp_ref.square_box_ref([3, 7, 90, 200])
    assert b[2] - b[0] == b[3] - b[1]


def test_backproject_truncates_like_astype_uint():
    marks = np.array([[0.5, 0.25], [0.999, 0.0]], np.float32)
    out = warp_ref.backproject_marks_ref(marks, [10, 20, 110, 120])
    assert out.dtype == np.uint and out.tolist() == [[60, 45], [109, 20]]


def test_crop_resize_identity_and_halving():
    rng = np.random.default_rng(2)
    frame = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    same = warp_ref.crop_resize_ref(frame, np.array([[4, 2, 14, 12]]), 10, 10)
    np.testing.assert_array_equal(same[0], frame[2:12, 4:14])
    half = warp_ref.crop_resize_ref(frame, np.array([[0, 0, 20, 20]]), 10, 10)
    exp = np.rint(frame[:20, :20].astype(np.float32).reshape(10, 2, 10, 2, 3).mean((1, 3)))
    assert np.abs(half[0].astype(np.float32) - exp).max() <= 1
