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
