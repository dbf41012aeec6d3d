"""Landmark mode without the probability tensor (flm_convt.hip, epilogue 3) against the path that
materialises [N,264,264,68] and decodes it: the two must agree bit for bit (float64 coordinates),
because the candidate lists hold every pixel that can enter the top n and the final selection uses
the same (value, flat index) keys.  Also covered: the overflow fallback (lists too small; flat maps
where every pixel ties with the threshold) and rejected landmarks (thresh above some of the means)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def flm():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import flm_amd
    return flm_amd


@pytest.fixture(scope="module")
def weights68():
    from flm_amd.weights import synth_fcn8_weights
    return synth_fcn8_weights(68, seed=2)


def _landmarks(model, xd, n_points, thresh, candidates, cap_div=1):
    """The options that pick the path are per-call arguments (flm_forward_opts), so each layout has its own workspace."""
    opts = dict(landmark_candidates=1 if candidates else 0, candidate_cap_div=cap_div)
    return model.forward_device(xd, "landmarks", n_points=n_points, thresh=thresh, opts=opts).cpu().numpy()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_candidate_path_equals_materialised_decode(flm, weights68, dtype):
    from flm_amd.networks import LANDMARKS_MODELS
    rng = np.random.default_rng(41)
    n = 5
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dtype)
    model.load_weights(weights68)
    xd = torch.from_numpy(rng.integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).cuda()
    for n_points, thresh in ((1, 0.0), (9, 0.0), (25, 0.0), (32, 0.0), (4, 0.0), (4, 0.5)):
        ref = _landmarks(model, xd, n_points, thresh, candidates=False)
        got = _landmarks(model, xd, n_points, thresh, candidates=True)
        assert got.shape == (n, 68, 2) and got.dtype == np.float64
        assert np.array_equal(got, ref), (dtype, n_points, thresh, np.abs(got - ref).max())
    assert (ref == -1).any() and (ref != -1).any()  # thresh 0.5 rejects part of the landmarks: both branches ran


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_eight_wave_kernel_equals_generic_kernel(flm, weights68, dtype):
    """flm_set_tuning "up3_cand8": the 8-wave candidate kernel (bf16; the knob's fp32 bit is accepted and ignored since
    round 3, so the fp32 leg checks that it is harmless) against the generic
    kernel and against the materialised decode, n = 4 and 25, a ragged batch (faces end inside a workgroup) and forced
    rows per workgroup; also with lists shrunk until they overflow (the gated fallback must still give the exact result)."""
    from flm_amd import _lib
    from flm_amd.networks import LANDMARKS_MODELS
    lib = _lib.load()
    rng = np.random.default_rng(47)
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dtype)
    model.load_weights(weights68)
    xd = torch.from_numpy(rng.integers(0, 256, (9, 256, 256, 3), dtype=np.uint8)).cuda()
    try:
        for n_points in (4, 25):
            ref = _landmarks(model, xd, n_points, 0.0, candidates=False)
            for knob, rows in ((0, 0), (3, 0), (3, 1), (3, 8)):
                _lib.check(lib.flm_set_tuning(b"up3_cand8", knob), "set_tuning")
                _lib.check(lib.flm_set_tuning(b"up3_cand8_rows", rows), "set_tuning")
                got = _landmarks(model, xd, n_points, 0.0, candidates=True)
                assert np.array_equal(got, ref), (dtype, n_points, knob, rows)
        _lib.check(lib.flm_set_tuning(b"up3_cand8", 3), "set_tuning")
        got = _landmarks(model, xd, 4, 0.0, candidates=True, cap_div=4096)
        assert np.array_equal(got, _landmarks(model, xd, 4, 0.0, candidates=False))
    finally:
        _lib.check(lib.flm_set_tuning(b"up3_cand8", 1), "set_tuning")
        _lib.check(lib.flm_set_tuning(b"up3_cand8_rows", 0), "set_tuning")


def test_weights_in_registers_kernel_equals_materialised_decode(flm, weights68):
    """flm_set_tuning "up3_wreg" = 1: up3_wreg_kernel (flm_up3_wreg.hip; bf16 only) against the materialised decode, bit for
    bit: 256 x 256 (four bands of position rows per face) at 1, 9 and 40 faces (40: more faces than face chunks, so a
    workgroup walks two), n = 4 and 25; a non-square input whose face fits one band; lists shrunk until they overflow (the
    gated fallback must still give the exact result)."""
    from flm_amd import _lib
    from flm_amd.networks import LANDMARKS_MODELS
    lib = _lib.load()
    rng = np.random.default_rng(48)
    try:
        _lib.check(lib.flm_set_tuning(b"up3_wreg", 1), "set_tuning")
        model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype="bf16")
        model.load_weights(weights68)
        for n in (1, 9, 40):
            xd = torch.from_numpy(rng.integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).cuda()
            for n_points in (4, 25):
                ref = _landmarks(model, xd, n_points, 0.0, candidates=False)
                got = _landmarks(model, xd, n_points, 0.0, candidates=True)
                assert np.array_equal(got, ref), (n, n_points, np.abs(got - ref).max())
            _lib.check(lib.flm_set_tuning(b"up3_wreg", 0), "set_tuning")
            other = _landmarks(model, xd, 25, 0.0, candidates=True)     # the 8-wave kernel on the same batch
            _lib.check(lib.flm_set_tuning(b"up3_wreg", 1), "set_tuning")
            assert np.array_equal(other, ref)
        got = _landmarks(model, xd, 4, 0.0, candidates=True, cap_div=4096)
        assert np.array_equal(got, _landmarks(model, xd, 4, 0.0, candidates=False))
        # flat maps (zero upsampling kernels: p = 1/68 everywhere, every lane of every tile stores a record, every list
        # overflows): the record lists flush once per tile and the gated fallback must give the tie rule's result
        wz = {k: np.array(v, copy=True) for k, v in weights68.items()}
        for k in list(wz):
            if k.startswith(("up3", "up4", "up5")):
                wz[k][...] = 0
        flat = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype="bf16")
        flat.load_weights(wz)
        xd = torch.from_numpy(rng.integers(0, 256, (3, 256, 256, 3), dtype=np.uint8)).cuda()
        assert np.array_equal(_landmarks(flat, xd, 4, 0.0, candidates=True), _landmarks(flat, xd, 4, 0.0, candidates=False))
        model = LANDMARKS_MODELS["fcn_8"](68, input_height=96, input_width=160, dtype="bf16")
        model.load_weights(weights68)
        xd = torch.from_numpy(rng.integers(0, 256, (35, 96, 160, 3), dtype=np.uint8)).cuda()
        for n_points in (4, 12):
            ref = _landmarks(model, xd, n_points, 0.0, candidates=False)
            got = _landmarks(model, xd, n_points, 0.0, candidates=True)
            assert np.array_equal(got, ref), ("96x160", n_points)
    finally:
        _lib.check(lib.flm_set_tuning(b"up3_wreg", 0), "set_tuning")


def test_candidate_overflow_falls_back(flm, weights68):
    """Lists 1/4096 of their size overflow in every face: the gated materialising launch must take over."""
    from flm_amd.networks import LANDMARKS_MODELS
    rng = np.random.default_rng(42)
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256)
    model.load_weights(weights68)
    xd = torch.from_numpy(rng.integers(0, 256, (3, 256, 256, 3), dtype=np.uint8)).cuda()
    ref = _landmarks(model, xd, 4, 0.0, candidates=False)
    for cap_div in (4096, 0x7fffffff):     # the second one would size an empty list: the library keeps one 64-key block
        got = _landmarks(model, xd, 4, 0.0, candidates=True, cap_div=cap_div)
        assert np.array_equal(got, ref), cap_div


def test_candidate_flat_maps(flm, weights68):
    """Zero score / upsampling kernels: every pixel has p = 1/68 exactly, all tie with the threshold, every list
    overflows; the reference's tie rule (largest flat indices) must still come out."""
    from flm_amd.networks import LANDMARKS_MODELS
    from oracle import decode_ref
    w = {k: np.array(v, copy=True) for k, v in weights68.items()}
    for k in list(w):
        if k.startswith(("up3", "up4", "up5")):
            w[k][...] = 0
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256)
    model.load_weights(w)
    xd = torch.from_numpy(np.random.default_rng(43).integers(0, 256, (2, 256, 256, 3), dtype=np.uint8)).cuda()
    ref = _landmarks(model, xd, 4, 0.0, candidates=False)
    got = _landmarks(model, xd, 4, 0.0, candidates=True)
    assert np.array_equal(got, ref)
    flat = np.full((1, 264, 264, 68), np.float32(1.0) / np.float32(68.0), dtype=np.float32)
    exp = decode_ref.transfer_target_ref(flat, 0, 4).reshape(68, 2)
    assert np.allclose(got[0], exp, rtol=0, atol=1e-9)


def test_candidate_path_batch_of_one_and_ragged_batch(flm, weights68):
    from flm_amd.networks import LANDMARKS_MODELS
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256)
    model.load_weights(weights68)
    rng = np.random.default_rng(44)
    for n in (1, 7):
        xd = torch.from_numpy(rng.integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).cuda()
        ref = _landmarks(model, xd, 4, 0.0, candidates=False)
        got = _landmarks(model, xd, 4, 0.0, candidates=True)
        assert np.array_equal(got, ref), n


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_candidate_path_on_a_non_square_input(flm, weights68, dtype):
    """96x160 crops -> 104x168 maps: 13x21 input positions per face do not fill the position tiles, faces are padded
    per workgroup, and the sampled phases come from partly empty tiles."""
    from flm_amd.networks import LANDMARKS_MODELS
    rng = np.random.default_rng(45)
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=96, input_width=160, dtype=dtype)
    model.load_weights(weights68)
    xd = torch.from_numpy(rng.integers(0, 256, (3, 96, 160, 3), dtype=np.uint8)).cuda()
    for n_points in (4, 12):
        ref = _landmarks(model, xd, n_points, 0.0, candidates=False)
        got = _landmarks(model, xd, n_points, 0.0, candidates=True)
        assert np.array_equal(got, ref), (dtype, n_points)
    assert ref[..., 0].max() < 168 and ref[..., 1].max() < 104
    # 130 faces: the key merge switches from one class per wave (below 128 faces) to five (flm_decode.hip)
    xd = torch.from_numpy(rng.integers(0, 256, (130, 96, 160, 3), dtype=np.uint8)).cuda()
    ref = _landmarks(model, xd, 4, 0.0, candidates=False)
    got = _landmarks(model, xd, 4, 0.0, candidates=True)
    assert np.array_equal(got, ref), (dtype, "130 faces")
    # 200 faces: from 192 the thresholds come from the coalesced kernel (lists of 4 or 8 in registers, flm_decode.hip
    # cand_tau_small_kernel); n = 3 and 4 take its 4-entry lists, 7 and 8 the 8-entry ones, 9 the wave-per-class kernel
    xd = torch.from_numpy(rng.integers(0, 256, (200, 96, 160, 3), dtype=np.uint8)).cuda()
    for n_points in (3, 4, 7, 8, 9):
        ref = _landmarks(model, xd, n_points, 0.0, candidates=False)
        got = _landmarks(model, xd, n_points, 0.0, candidates=True)
        assert np.array_equal(got, ref), (dtype, "200 faces", n_points)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_candidate_path_does_not_need_its_fallback_on_ordinary_maps(flm, weights68, dtype):
    """The gated materialising launch keeps results exact when a list overflows or a class ends short of n keys, but it
    doubles the cost: on the bench's maps the flag must stay clear and the lists far from full (a threshold taken from
    probabilities that differ from the main launch's would trip the "fewer than n keys" check)."""
    from flm_amd import _lib
    from flm_amd.networks import LANDMARKS_MODELS
    lib = _lib.load()
    n = 6
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dtype)
    model.load_weights(weights68)
    xd = torch.from_numpy(np.random.default_rng(46).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).cuda()
    for n_points in (4, 25):
        ws = model.new_workspace(n, "landmarks", n_points)
        model.forward_device(xd, "landmarks", n_points=n_points, workspace=ws)
        torch.cuda.synchronize()

        def off(name):
            return lib.flm_fcn8_workspace_offset(name, n, 256, 256, 68, model._dt, _lib.OUT_LANDMARKS,
                                                 _lib.DECODE_TOPN, n_points)
        cap = off(b"cand_cap")
        cnt = ws[off(b"cand_cnt"):off(b"cand_cnt") + 4 * (n + 1)].view(torch.int32).cpu().numpy()
        assert cnt[n] == 0, (dtype, n_points, "fallback flag raised")
        assert 68 * n_points <= cnt[:n].min() and cnt[:n].max() < cap // 2, (dtype, n_points, cnt[:n], cap)
