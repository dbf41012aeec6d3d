"""Pin oracle/decode_ref.py against the reference's own utils/metrics.py.

Golden fixture: tests/golden/decode_golden.npz, produced by
tests/golden/make_decode_golden.py which RUNS the reference implementation
(utils/metrics.py:46-109) in the build container.
"""
import importlib.util
import os

import numpy as np
import pytest

from oracle import decode_ref

REF_METRICS = "/root/reference/keypoints_detector/utils/metrics.py"


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "decode_golden.npz"))


def test_get_average_xy_matches_golden(gold):
    modes = gold["modes"]
    ths = gold["thresholds"]
    for name in gold["names"]:
        hm = gold["hm_" + str(name)]
        exp = gold["xy_" + str(name)]
        for i, n in enumerate(modes):
            for j, t in enumerate(ths):
                with np.errstate(all="ignore"):
                    got = decode_ref.get_average_xy_ref(hm, int(n), float(t) if t else 0)
                got = np.array([float(got[0]), float(got[1])])
                # same numpy ops in the same order -> bit-identical
                assert np.array_equal(got, exp[i, j]), (name, n, t, got, exp[i, j])


def test_transfer_target_as_shipped(gold):
    y = gold["tt_input"]
    with np.errstate(all="ignore"):
        a = decode_ref.transfer_target_ref(y, as_shipped=True)
        b = decode_ref.transfer_target_ref(y, 0.2, 25, as_shipped=True)
    assert np.array_equal(a, gold["tt_shipped_default"])
    # the positional slip makes the arguments irrelevant (utils/metrics.py:98)
    assert np.array_equal(b, gold["tt_shipped_args"])
    assert np.array_equal(gold["tt_shipped_default"], gold["tt_shipped_args"])
    # the all-zero landmark map is rejected as (-1,-1)
    assert a[1, 4] == -1 and a[1, 5] == -1
    # and the fixed-argument form equals as-shipped when asked for n=4, thresh=0
    with np.errstate(all="ignore"):
        c = decode_ref.transfer_target_ref(y, 0, 4)
    assert np.array_equal(a, c)


@pytest.mark.skipif(not os.path.isfile(REF_METRICS), reason="reference not present (GPU box)")
def test_live_reference_property():
    """Seeded sweep against the live import of the reference (container only)."""
    os.environ.setdefault("MPLBACKEND", "Agg")
    spec = importlib.util.spec_from_file_location("ref_metrics_live", REF_METRICS)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(7)
    for k in range(40):
        h, w = int(rng.integers(4, 70)), int(rng.integers(4, 70))
        hm = rng.random((h, w), dtype=np.float32)
        if k % 5 == 0:
            hm = hm.astype(np.float64)
        n = int(rng.choice([0, 1, 2, 4, 7, 16]))
        n = min(n, h * w)
        t = float(rng.choice([0.0, 0.3, 0.6]))
        with np.errstate(all="ignore"):
            exp = ref.get_average_xy(hm, h, w, n, t)
            got = decode_ref.get_average_xy_ref(hm, n, t)
        assert float(exp[0]) == float(got[0]) and float(exp[1]) == float(got[1]), (k, h, w, n, t)


def test_topn_gap_rel_is_the_relative_gap_at_the_nth_place():
    """decode_ref.topn_gap_rel (the checker's "determined pair" criterion, used by bench.py and the config tests): against a
    full sort, with ties and with n + 1 = the whole map."""
    rng = np.random.default_rng(21)
    m = rng.random((500, 7)).astype(np.float64)
    m[10, 3] = m[20, 3] = m[:, 3].max() + 1.0            # a tie at the top
    for n in (1, 4, 25, 499):
        srt = np.sort(m, axis=0)[::-1]
        exp = (srt[n - 1] - srt[n]) / srt[n - 1]
        np.testing.assert_array_equal(decode_ref.topn_gap_rel(m, n), exp)
    assert decode_ref.topn_gap_rel(m, 1)[3] == 0.0        # the tie: gap zero -> undetermined for n = 1
