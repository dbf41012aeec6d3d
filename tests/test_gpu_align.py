"""GPU parity: similarity estimate, alignment warp and crop front-end vs oracle/warp_ref.py.

Bars (BASELINE.json north_star): warped pixels within 1 ULP of the CPU restatement (the
restatement emulates fma through float64, so a double rounding may move a value by one ULP).
"""
import os

import numpy as np
import pytest
import torch

from oracle import warp_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import flm_amd  # noqa: F401
    from flm_amd import alignment, _lib
    _lib.load()
    return alignment


def ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7fffffff), ai)
    bi = np.where(bi < 0, -(bi & 0x7fffffff), bi)
    return np.abs(ai - bi)


def rand_sim(rng, n):
    m = np.zeros((n, 2, 3), np.float32)
    for i in range(n):
        s, th = rng.uniform(0.5, 1.8), rng.uniform(-0.8, 0.8)
        m[i] = [[s * np.cos(th), -s * np.sin(th), rng.uniform(-20, 20)],
                [s * np.sin(th), s * np.cos(th), rng.uniform(-20, 20)]]
    return m


@pytest.mark.parametrize("u8", [True, False])
def test_warp_within_one_ulp(A, u8):
    rng = np.random.default_rng(11)
    n, hs, ws, hd, wd = 5, 61, 77, 48, 52
    src = rng.integers(0, 256, (n, hs, ws, 3), dtype=np.uint8)
    if not u8:
        src = (src.astype(np.float32) - 117.3) * 1.37
    m = rand_sim(rng, n)
    exp = warp_ref.warp_affine_ref(src, m, hd, wd)
    got = A.warp_device(torch.from_numpy(src).cuda(), torch.from_numpy(m).cuda(), hd, wd).cpu().numpy()
    assert got.shape == exp.shape
    # 1 ULP, or an absolute 1e-5 on results that cancel to (near) zero
    bad = (ulp_diff(got, exp) > 1) & (np.abs(got - exp) > 1e-5)
    assert not bad.any(), (int(bad.sum()), float(np.abs(got - exp).max()))


def test_identity_warp_exact(A):
    rng = np.random.default_rng(12)
    src = rng.integers(0, 256, (3, 32, 40, 3), dtype=np.uint8)
    m = np.tile(np.array([[1, 0, 0], [0, 1, 0]], np.float32), (3, 1, 1))
    got = A.warp_device(torch.from_numpy(src).cuda(), torch.from_numpy(m).cuda(), 32, 40).cpu().numpy()
    assert np.array_equal(got, src.astype(np.float32))


def test_similarity_vs_oracle_and_skimage(A, golden_dir):
    gold = np.load(os.path.join(golden_dir, "warp_golden.npz"))
    rng = np.random.default_rng(13)
    lm = rng.uniform(0, 263, (7, 68, 2))
    lm[2, 5] = [-1, -1]
    lm[3, :] = -1            # every landmark rejected -> identity
    tm = A.canonical_template(68, 256, 256)
    exp = warp_ref.similarity_ref(lm, tm)
    got = A.similarity_device(torch.from_numpy(lm).cuda(), torch.from_numpy(tm).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, exp, rtol=1e-6, atol=1e-6)
    assert np.array_equal(got[3], np.array([[1, 0, 0], [0, 1, 0]], np.float32))
    # grid-to-crop scale applied inside the kernel (float64 products, reject marker kept): the same bits as scaling first
    sc = (256.0 / 264.0, 250.0 / 264.0)
    pre = np.where(lm < 0, lm, lm * np.array(sc))
    a = A.similarity_device(torch.from_numpy(pre).cuda(), torch.from_numpy(tm).cuda()).cpu().numpy()
    b = A.similarity_device(torch.from_numpy(lm).cuda(), torch.from_numpy(tm).cuda(), sc).cpu().numpy()
    assert np.array_equal(a, b)
    np.testing.assert_allclose(b, warp_ref.similarity_ref(pre, tm), rtol=1e-6, atol=1e-6)
    for i in range(4):   # scikit-image's Umeyama estimate on its own points
        g = A.similarity_device(torch.from_numpy(gold["src_pts"][i:i + 1]).cuda(),
                                torch.from_numpy(gold["dst_pts"][i]).cuda()).cpu().numpy()[0]
        np.testing.assert_allclose(g, gold["mats"][i][:2], rtol=2e-6, atol=2e-5)


def test_warp_vs_skimage_golden(A, golden_dir):
    gold = np.load(os.path.join(golden_dir, "warp_golden.npz"))
    m = torch.from_numpy(gold["mats"][:, :2, :].astype(np.float32)).cuda()
    got = A.warp_device(torch.from_numpy(gold["imgs"]).cuda(), m, 40, 44).cpu().numpy()
    assert np.abs(got - gold["warped"]).max() < 0.05


def test_crop_resize_vs_oracle():
    import flm_amd  # noqa: F401
    from flm_amd import prediction
    rng = np.random.default_rng(14)
    frame = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    faces = [[30, 40, 130, 150], [200, 20, 260, 140], [-10, 100, 90, 260], [400, 180, 479, 269]]
    boxes = prediction.face_boxes(faces)
    assert boxes == [warp_ref.square_box_ref(f) for f in faces]
    got = prediction.crop_faces_device(torch.from_numpy(frame).cuda(), boxes, 64, 64).cpu().numpy()
    exp = warp_ref.crop_resize_ref(frame, np.asarray(boxes), 64, 64)
    d = np.abs(got.astype(np.int32) - exp.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


def test_get_image_array_device_matches_reference_semantics():
    import flm_amd  # noqa: F401
    from flm_amd.data.generator import get_image_array, DataLoaderError
    from oracle import fcn_ref
    rng = np.random.default_rng(15)
    img = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    x = get_image_array(img, 48, 32, ordering="channels_last")
    assert np.array_equal(x, fcn_ref.get_image_array_ref(img))
    xf = get_image_array(img, 48, 32)  # default ordering is channels_first (generator.py:33,67-68)
    assert xf.shape == (3, 32, 48) and np.array_equal(xf, np.rollaxis(x, 2, 0))
    d = get_image_array(img, 48, 32, imgNorm="divide", ordering="channels_last")
    assert np.array_equal(d, img.astype(np.float32) / 255.0)
    s = get_image_array(img, 48, 32, imgNorm="sub_and_divide", ordering="channels_last")
    assert np.array_equal(s, np.float32(img) / 127.5 - 1)
    with pytest.raises(DataLoaderError):
        get_image_array("/nonexistent/file.png", 48, 32)
    with pytest.raises(DataLoaderError):
        get_image_array(12345, 48, 32)
