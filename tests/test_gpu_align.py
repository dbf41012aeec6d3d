"""GPU parity: similarity estimate, alignment warp and crop front-end vs oracle/warp_ref.py.

Bars (BASELINE.json north_star): warped pixels within 1 ULP of the CPU restatement -- whose fma is correctly
rounded (oracle/warp_ref.py), so no escape for small values is needed; the similarity fit (float64, sequential sums
on both sides) and the crop/resize (integer fixed point on both sides) are bit-exact.
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
    u = ulp_diff(got, exp)
    print("warp (u8=%s): %d of %d values differ from the restatement, max %d ULP" % (u8, int((u > 0).sum()), u.size, int(u.max())))
    assert u.max() <= 1, (int((u > 1).sum()), float(np.abs(got - exp).max()))


@pytest.mark.parametrize("shape", [(3, 61, 77, 37, 64), (2, 40, 2, 20, 128), (2, 50, 90, 33, 320), (4, 256, 256, 256, 256)])
def test_warp_row_segment_kernel(A, shape):
    """Destination widths that are multiples of 64 run warp_u8_rows_kernel (a wave per 64-pixel row segment, a workgroup
    per strip of up to 256 columns; rows past the image, the x0 = ws - 1 column through the weight): within 1 ULP of the restatement like every warp, and the same
    bits as the pixel-list kernel (flm_set_tuning "warp_rows" 0) for every rows-per-wave form."""
    from flm_amd import _lib
    lib = _lib.load()
    n, hs, ws, hd, wd = shape
    rng = np.random.default_rng(13)
    src = rng.integers(0, 256, (n, hs, ws, 3), dtype=np.uint8)
    m = rand_sim(rng, n)
    m[0] = [[1, 0, 0], [0, 1, 0]]            # identity: the last column is xs = ws - 1 exactly when wd >= ws
    exp = warp_ref.warp_affine_ref(src, m, hd, wd)
    got = {}
    try:
        for knob in (0, 1, 4):
            _lib.check(lib.flm_set_tuning(b"warp_rows", knob), "set_tuning")
            got[knob] = A.warp_device(torch.from_numpy(src).cuda(), torch.from_numpy(m).cuda(), hd, wd).cpu().numpy()
    finally:
        _lib.check(lib.flm_set_tuning(b"warp_rows", 1), "set_tuning")
    assert ulp_diff(got[1], exp).max() <= 1
    for knob in (1, 4):
        assert np.array_equal(got[0], got[knob]), knob
    assert lib.flm_set_tuning(b"warp_rows", 3) != 0


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
    assert np.array_equal(got, exp)         # float64 sums in landmark order on both sides, rounded to float32 once
    assert np.array_equal(got[3], np.array([[1, 0, 0], [0, 1, 0]], np.float32))
    # grid-to-crop scale applied inside the kernel (float64 products, reject marker kept): the same bits as scaling first
    sc = (256.0 / 264.0, 250.0 / 264.0)
    pre = np.where(lm < 0, lm, lm * np.array(sc))
    a = A.similarity_device(torch.from_numpy(pre).cuda(), torch.from_numpy(tm).cuda()).cpu().numpy()
    b = A.similarity_device(torch.from_numpy(lm).cuda(), torch.from_numpy(tm).cuda(), sc).cpu().numpy()
    assert np.array_equal(a, b)
    assert np.array_equal(b, warp_ref.similarity_ref(pre, tm))
    for i in range(4):   # scikit-image's Umeyama estimate on its own points
        g = A.similarity_device(torch.from_numpy(gold["src_pts"][i:i + 1]).cuda(),
                                torch.from_numpy(gold["dst_pts"][i]).cuda()).cpu().numpy()[0]
        np.testing.assert_allclose(g, gold["mats"][i][:2], rtol=2e-6, atol=2e-5)


def test_warp_vs_skimage_golden(A, golden_dir):
    gold = np.load(os.path.join(golden_dir, "warp_golden.npz"))
    m = torch.from_numpy(gold["mats"][:, :2, :].astype(np.float32)).cuda()
    got = A.warp_device(torch.from_numpy(gold["imgs"]).cuda(), m, 40, 44).cpu().numpy()
    # skimage maps coordinates in float64, the spec in float32: three fmas on coordinates below 64 px leave at most
    # ~3 * 2^-24 * 64 = 1.1e-5 px, times a gradient of at most 255 grey levels per px along each axis
    d = np.abs(got - gold["warped"]).max()
    print("warp vs scikit-image: max %.3g grey levels" % d)
    assert d < 2 * 255 * 1.2e-5


def test_crop_resize_vs_oracle():
    """uint8 -> uint8 in integer fixed point on both sides (OpenCV's 8-bit INTER_LINEAR restated): bit-exact.  Boxes
    inside the frame, poking out of every side, outside it, up- and down-scaling, the exact-2x area path."""
    import flm_amd  # noqa: F401
    from flm_amd import prediction
    from flm_amd.data.generator import resize_u8_device
    rng = np.random.default_rng(14)
    frame = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    faces = [[30, 40, 130, 150], [200, 20, 260, 140], [-10, 100, 90, 260], [400, 180, 479, 269], [5, 5, 25, 30]]
    boxes = prediction.face_boxes(faces)
    assert boxes == [warp_ref.square_box_ref(f) for f in faces]
    boxes = boxes + [[100, 60, 228, 188], [-40, -40, 30, 30], [450, 240, 520, 310], [600, 10, 700, 110], [0, 0, 480, 270]]
    fd = torch.from_numpy(frame).cuda()
    for oh, ow in ((64, 64), (128, 96), (33, 47)):
        got = prediction.crop_faces_device(fd, boxes, oh, ow).cpu().numpy()
        exp = warp_ref.crop_resize_ref(frame, np.asarray(boxes), oh, ow)
        assert np.array_equal(got, exp), (oh, ow, int((got != exp).sum()))
    assert not got[8].any()                       # box beyond the frame: zeros
    # [100,60,228,188] -> 64x64 is the exact 2x downscale: cv2 averages 2x2 blocks there
    blk = frame[60:188, 100:228].astype(np.int32)
    area = (blk[0::2, 0::2] + blk[0::2, 1::2] + blk[1::2, 0::2] + blk[1::2, 1::2] + 2) >> 2
    got = prediction.crop_faces_device(fd, boxes, 64, 64).cpu().numpy()
    assert np.array_equal(got[5], area.astype(np.uint8))
    # whole-image resize of get_image_array (data/generator.py:53)
    for oh, ow in ((256, 256), (135, 240), (540, 960)):
        got = resize_u8_device(fd, oh, ow).cpu().numpy()
        assert np.array_equal(got, warp_ref.resize_u8_ref(frame, oh, ow)), (oh, ow)


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
