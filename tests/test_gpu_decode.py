"""GPU parity: HIP decode (flm_decode through the C ABI) vs the golden vectors produced by the
reference's own utils/metrics.py and vs oracle/decode_ref.py.

Bars: top-n mode is BIT-EXACT (same selection, same float32/float64 operation order as the
reference, utils/metrics.py:66-80).  All-pixel mode agrees to 1e-4 px: the reference's hsum is
numpy's float32 pairwise sum (:60), whose rounding cannot be reproduced bit for bit; the device
sums in float64 and rounds once.
"""
import os

import numpy as np
import pytest
import torch

from oracle import decode_ref

pytestmark = pytest.mark.gpu
ALL_TOL = 1e-4


@pytest.fixture(scope="module")
def M():
    import flm_amd  # noqa: F401
    from flm_amd.utils import metrics
    from flm_amd import _lib
    _lib.load()
    return metrics


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "decode_golden.npz"))


def test_golden_get_average_xy(M, gold):
    for name in gold["names"]:
        hm = gold["hm_" + str(name)]
        exp = gold["xy_" + str(name)]
        for i, n in enumerate(gold["modes"]):
            for j, t in enumerate(gold["thresholds"]):
                got = np.array(M.get_average_xy(hm, hm.shape[0], hm.shape[1], int(n), float(t)))
                e = exp[i, j]
                if int(n) >= 1:
                    assert np.array_equal(got, e), (name, n, t, got, e)
                else:
                    assert np.abs(got - e).max() <= ALL_TOL, (name, n, t, got, e)


def test_golden_transfer_target_as_shipped(M, gold):
    y = gold["tt_input"]
    assert np.array_equal(M.transfer_target(y, as_shipped=True), gold["tt_shipped_default"])
    assert np.array_equal(M.transfer_target(y), gold["tt_shipped_default"])   # the default call IS the reference's
    assert np.array_equal(M.transfer_target(y, 0.2, 25, as_shipped=True), gold["tt_shipped_args"])
    # honouring the documented arguments instead (n=4, thresh=0 asked explicitly) gives the same
    assert np.array_equal(M.transfer_target(y, 0, 4), gold["tt_shipped_default"])
    lst = M.transfer_xy_coord(y[0], 4, 0)
    assert isinstance(lst, list) and len(lst) == 10
    assert np.array_equal(np.array(lst), gold["tt_shipped_default"][0])


@pytest.mark.parametrize("shape", [(3, 48, 56, 5), (2, 33, 17, 68), (1, 7, 9, 96), (5, 64, 64, 21)])
def test_random_vs_oracle(M, shape):
    rng = np.random.default_rng(sum(shape))
    y = rng.random(shape, dtype=np.float32)
    y[0, :, :, 1] = 0
    for n, t in [(1, 0), (4, 0), (9, 0.5), (25, 0.2), (64, 0), (0, 0), (0, 0.6)]:
        if n > shape[1] * shape[2]:
            continue
        with np.errstate(all="ignore"):
            exp = decode_ref.transfer_target_ref(y, t, n)
        got = M.transfer_target(y, t, n)
        if n >= 1:
            assert np.array_equal(got, exp), (shape, n, t)
        else:
            assert np.abs(got - exp).max() <= ALL_TOL


def test_ties_follow_value_then_index_rule(M):
    hm = np.zeros((1, 12, 10, 2), np.float32)
    hm[0, 2:6, 3:7, 0] = 0.75          # a 16-pixel plateau of exactly equal maxima
    hm[0, :, :, 1] = 1.0               # a completely flat map
    for n in (1, 3, 4, 7, 16, 20):
        exp = decode_ref.transfer_target_ref(hm, 0, n)   # stable sort = (value, index) order
        got = M.transfer_target(hm, 0, n)
        assert np.array_equal(got, exp), n


def test_negative_and_tiny_maps(M):
    rng = np.random.default_rng(3)
    y = (rng.standard_normal((2, 5, 6, 3))).astype(np.float32)   # negative values order correctly
    for n in (1, 4, 30):
        with np.errstate(all="ignore"):
            exp = decode_ref.transfer_target_ref(y, -10.0, n)
        got = M.transfer_target(y, -10.0, n)
        assert np.array_equal(got, exp, equal_nan=True)


def test_wide_lists_n_up_to_128(M):
    """n_points beyond 64 (the reference's sweep calls transfer_target with n = k*k up to 81, utils/metrics.py:130-133):
    two list registers per lane; bit-equal to the oracle, ties included, for maps of 68 and of 80 channels (the
    17- and 24-channels-per-wave kernels), a map smaller than n, and the forward's landmark mode."""
    rng = np.random.default_rng(9)
    for shape in ((2, 40, 37, 68), (1, 24, 24, 80), (1, 8, 9, 3)):
        y = rng.random(shape, dtype=np.float32)
        y[0, :3, :5, 0] = 0.75          # ties at and around the n-th place
        for n in (65, 81, 100, 128):
            with np.errstate(all="ignore"):
                exp = decode_ref.transfer_target_ref(y, 0.2, n)
            got = M.transfer_target(y, 0.2, n)
            assert np.array_equal(got, exp, equal_nan=True), (shape, n)


def test_lds_dma_form_equals_register_prefetch_form(M):
    """flm_set_tuning "decode_lds_dma": the LDS-DMA ring (68-landmark maps, the default) against the register-prefetch
    kernel -- same coordinates bit for bit in top-n and all-pixel mode, on a map whose chunks end inside a tile (the
    zero fill of the buffer bounds check) and at the headline size."""
    from flm_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(5)
    for shape in ((3, 50, 37, 68), (64, 264, 264, 68)):
        hm = torch.rand(shape, device="cuda", generator=g)
        hm[0, :2, :7, 3] = 0.875      # ties
        out = {}
        try:
            for knob in (0, 1):
                _lib.check(lib.flm_set_tuning(b"decode_lds_dma", knob), "set_tuning")
                out[knob] = [M.decode_device(hm, n, 0.1).cpu().numpy() for n in (0, 1, 4, 25, 64)]
        finally:
            _lib.check(lib.flm_set_tuning(b"decode_lds_dma", 1), "set_tuning")
        for a, b, n in zip(out[0], out[1], (0, 1, 4, 25, 64)):
            assert np.array_equal(a, b, equal_nan=True), (shape, n)
        del hm


def test_unsupported_n_points_raises(M):
    from flm_amd._lib import FlmError
    with pytest.raises(FlmError):
        M.transfer_target(np.zeros((1, 16, 16, 2), np.float32), 0, 129)


def test_full_size_properties(M):
    """BASELINE config-2 size (64 x 264 x 264 x 68): size-independent properties."""
    n, h, w, l = 64, 264, 264, 68
    g = torch.Generator(device="cuda").manual_seed(1)
    hm = torch.rand((n, h, w, l), device="cuda", generator=g)
    a = M.decode_device(hm, 4, 0.0)
    # (1) scaling by a power of two is exact in fp: identical coordinates, bit for bit
    b = M.decode_device(hm * 0.25, 4, 0.0)
    assert torch.equal(a, b)
    # (2) a channel permutation permutes the outputs
    perm = torch.randperm(l, device="cuda", generator=g)
    c = M.decode_device(hm[..., perm].contiguous(), 4, 0.0)
    assert torch.equal(c, a[:, perm])
    # (3) a one-hot map decodes to its pixel exactly, in both modes
    oh = torch.zeros((n, h, w, l), device="cuda")
    ys = torch.randint(0, h, (n, l), device="cuda", generator=g)
    xs = torch.randint(0, w, (n, l), device="cuda", generator=g)
    ni = torch.arange(n, device="cuda")[:, None].expand(n, l)
    li = torch.arange(l, device="cuda")[None, :].expand(n, l)
    oh[ni, ys, xs, li] = 0.5
    for npts in (1, 4, 0):
        d = M.decode_device(oh, npts, 0.0)
        assert torch.equal(d[..., 0], xs.double()) and torch.equal(d[..., 1], ys.double())
    # (4) spot-check a few (face, landmark) pairs against the oracle at full size
    hm_np = hm[:2].cpu().numpy()
    for f, ch in [(0, 0), (1, 67), (0, 33)]:
        with np.errstate(all="ignore"):
            e4 = decode_ref.get_average_xy_ref(hm_np[f, :, :, ch], 4, 0)
            e0 = decode_ref.get_average_xy_ref(hm_np[f, :, :, ch], 0, 0)
        assert np.array_equal(a[f, ch].cpu().numpy(), np.array(e4, np.float64))
        d0 = M.decode_device(hm[:2], 0, 0.0)[f, ch].cpu().numpy()
        assert np.abs(d0 - np.array(e0, np.float64)).max() <= ALL_TOL
