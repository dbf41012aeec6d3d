"""GPU parity: HIP FCN-8 forward (through the C ABI) vs the CPU oracle (oracle/fcn_ref.py).

Tolerances (fp32 path): probabilities max-abs <= 1e-5, intermediates relative 2e-5 of the
tensor's max magnitude (both sides accumulate K up to 12544 products in fp32, in different
orders).  Parity of the forward against the reference itself is unpinned (TensorFlow absent,
no reference fixtures): the oracle is the build's restatement, see oracle/__init__.py.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def flm():
    import flm_amd
    from flm_amd import _lib
    _lib.load()
    return flm_amd


@pytest.fixture(scope="module")
def weights68():
    from flm_amd.weights import synth_fcn8_weights
    return synth_fcn8_weights(68, seed=2)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _run_case(flm, params, n, h, w, c, seed, u8):
    from flm_amd.networks import LANDMARKS_MODELS
    from oracle import fcn_ref
    rng = np.random.default_rng(seed)
    model = LANDMARKS_MODELS["fcn_8"](c, input_height=h, input_width=w)
    model.load_weights(params)
    if u8:
        img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        x_ref = np.stack([fcn_ref.get_image_array_ref(im) for im in img])
        xd = torch.from_numpy(img).cuda()
    else:
        x_ref = (rng.standard_normal((n, h, w, 3)) * 50).astype(np.float32)
        xd = torch.from_numpy(x_ref).cuda()
    logits_ref, inter = fcn_ref.fcn8_logits_ref(x_ref, params, torch.float32, return_intermediates=True)
    probs = model.forward_device(xd, "probs")
    torch.cuda.synchronize()
    for name in ("f1", "f2", "f3", "f4", "f5", "fc6", "fc7", "fuse4", "seg_feats"):
        got = model.intermediate(name, n, "probs").cpu().numpy()[..., : inter[name].shape[-1]]
        assert got.shape == inter[name].shape, name
        assert _rel(got, inter[name]) < 2e-5, (name, _rel(got, inter[name]))
    probs_ref = fcn_ref.fcn8_predict_ref(x_ref, params)
    got = probs.cpu().numpy()
    assert got.shape == probs_ref.shape == (n, (h + 8) * (w + 8), c)
    assert np.abs(got - probs_ref).max() <= 1e-5
    assert np.abs(got.sum(-1) - 1).max() < 1e-5
    if c % 4 == 0:
        lg = model.forward_device(xd, "logits").cpu().numpy()
        assert _rel(lg, logits_ref) < 2e-5
    # class map: equal to the oracle's argmax except where the oracle's top-2 are within rounding
    cm = model.forward_device(xd, "classmap").cpu().numpy()
    cm_ref = probs_ref.reshape(n, h + 8, w + 8, c).argmax(-1)
    diff = cm != cm_ref
    if diff.any():
        srt = np.sort(probs_ref.reshape(n, h + 8, w + 8, c), axis=-1)
        gap = srt[..., -1] - srt[..., -2]
        assert gap[diff].max() < 2e-6, "class map differs away from ties"
    assert diff.mean() < 1e-3
    return model, xd, probs_ref


def test_forward_small_c68_u8(flm, weights68):
    _run_case(flm, weights68, n=2, h=64, w=96, c=68, seed=1, u8=True)


def test_forward_small_c68_f32_single(flm, weights68):
    _run_case(flm, weights68, n=1, h=32, w=32, c=68, seed=3, u8=False)


def test_forward_256_full_resolution(flm, weights68):
    """BASELINE input size (256x256 -> 264x264x68), two faces: every intermediate, the probabilities,
    the class map, and the landmarks decoded from them.

    Landmark bar (BASELINE.json north_star): coordinates within 1e-4 px, NME <= 1e-4.  The yardstick
    is the float64 evaluation of the oracle; the float32 oracle is itself only an approximation of it,
    so where float32 arithmetic cannot resolve 1e-4 px (a top-n centroid divides by the sum of a few
    ~1e-2 probabilities) the HIP path must be no further from the float64 result than twice the
    float32 oracle is."""
    from flm_amd import prediction
    from oracle import decode_ref, fcn_ref
    model, xd, probs_ref = _run_case(flm, weights68, n=2, h=256, w=256, c=68, seed=7, u8=True)
    x_ref = np.stack([fcn_ref.get_image_array_ref(im) for im in xd.cpu().numpy()])
    probs64 = fcn_ref.fcn8_predict_ref(x_ref, weights68, torch.float64).reshape(2, 264, 264, 68)
    hm32 = probs_ref.reshape(2, 264, 264, 68)
    for npts in (0, 4, 25):
        lm = prediction.predict(xd, model, n_points=npts).cpu().numpy()
        with np.errstate(all="ignore"):
            e32 = decode_ref.transfer_target_ref(hm32, 0, npts).reshape(2, 68, 2)
            e64 = decode_ref.transfer_target_ref(probs64, 0, npts).reshape(2, 68, 2)
        if npts == 0:
            # all-pixel centroid is smooth in the heatmap
            assert np.abs(lm - e64).max() <= 1e-4, np.abs(lm - e64).max()
            assert np.abs(lm - e32).max() <= 1e-4
            assert np.linalg.norm(lm - e64, axis=-1).mean() / 256.0 <= 1e-4
            continue
        # top-n SELECTS pixels: where the n-th and (n+1)-th largest values are closer than the forward's
        # fp32 rounding the selection is not determined by the inputs -- compare the rest
        flat = probs64.reshape(2, -1, 68)
        srt = np.sort(flat, axis=1)
        gap = (srt[:, -npts, :] - srt[:, -npts - 1, :]) / srt[:, -npts, :]
        decided = gap > 2e-5
        assert decided.mean() > 0.5
        err_hip = np.abs(lm - e64)[decided]
        err_o32 = np.abs(e32 - e64)[decided]
        print("top-%d: max |hip-f64| %.3g px, max |oracle32-f64| %.3g px, NME %.3g" %
              (npts, err_hip.max(), err_o32.max(), np.linalg.norm((lm - e64), axis=-1)[decided].mean() / 256))
        assert err_hip.max() <= max(1e-4, 2 * err_o32.max()), (err_hip.max(), err_o32.max())
        assert np.linalg.norm((lm - e64), axis=-1)[decided].mean() / 256.0 <= 1e-4


def test_fp32_two_level_accumulation_beats_the_single_chain(flm, weights68):
    """The fp32 implicit GEMMs sum every 32-product k-step from zero and add the step sums into a second accumulator
    set (csrc/flm_igemm.hip, TWO).  Against the float64 oracle every layer must be closer than the single fmaf chain
    (knob f32_two_level = 0: round 2's kernel) and at least as close as the float32 CPU oracle's blocked sums."""
    from flm_amd import _lib
    from flm_amd.networks import LANDMARKS_MODELS
    from oracle import fcn_ref
    lib = _lib.load()
    n = 2
    crops = np.random.default_rng(1).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)
    x_ref = np.stack([fcn_ref.get_image_array_ref(im) for im in crops])
    _, i64 = fcn_ref.fcn8_logits_ref(x_ref, weights68, torch.float64, return_intermediates=True)
    _, i32 = fcn_ref.fcn8_logits_ref(x_ref, weights68, torch.float32, return_intermediates=True)
    rms = lambda a, b: float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256)
    model.load_weights(weights68)
    xd = torch.from_numpy(crops).cuda()
    names = ("f2", "f3", "f4", "f5", "fc6", "fc7")
    err = {}
    try:
        for two in (0, 1):
            _lib.check(lib.flm_set_tuning(b"f32_two_level", two), "set_tuning")
            model.forward_device(xd, "probs")
            torch.cuda.synchronize()
            err[two] = {k: rms(model.intermediate(k, n, "probs").cpu().numpy()[..., : i64[k].shape[-1]], i64[k]) for k in names}
    finally:
        _lib.check(lib.flm_set_tuning(b"f32_two_level", 1), "set_tuning")
    e32 = {k: rms(i32[k], i64[k]) for k in names}
    print("relative RMS error vs the float64 oracle:", {k: "%.2g / %.2g / %.2g" % (err[0][k], err[1][k], e32[k]) for k in names},
          "(single chain / two-level / float32 CPU oracle)")
    for k in names:
        assert err[1][k] < 0.6 * err[0][k], (k, err)
        assert err[1][k] <= 1.05 * e32[k], (k, err[1][k], e32[k])


@pytest.mark.parametrize("dtype,n", [("f32", 64), ("f32", 32), ("bf16", 64)])
def test_fc6_position_order_keeps_the_bits(flm, weights68, dtype, n):
    """At batches below a tile's rows the positions of fc6's 8x8 map that share a tile are chosen for their common filter
    taps (IgemmArgs::posperm, csrc/flm_igemm_args.h): only which tile computes a row changes, so fc6's output -- and
    everything after it -- must be the bits of the map-order launch (knob posmajor_order = 0).  f32 at 64 / 32 faces: two /
    four positions per 128-row tile; bf16: the 256-row kernel keeps map order (the knob must be a no-op there)."""
    from flm_amd import _lib
    from flm_amd.networks import LANDMARKS_MODELS
    lib = _lib.load()
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dtype)
    model.load_weights(weights68)
    xd = torch.from_numpy(np.random.default_rng(21).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).cuda()
    got = {}
    try:
        for knob in (0, 1):
            _lib.check(lib.flm_set_tuning(b"posmajor_order", knob), "set_tuning")
            lm = model.forward_device(xd, "landmarks", n_points=4).clone()
            model.forward_device(xd, "probs")
            torch.cuda.synchronize()
            got[knob] = (model.intermediate("fc6", n, "probs").clone(), lm)
    finally:
        _lib.check(lib.flm_set_tuning(b"posmajor_order", 1), "set_tuning")
    assert torch.equal(got[0][0], got[1][0]), "fc6"
    assert torch.equal(got[0][1], got[1][1]), "landmarks"


def test_fp32_batches_beyond_2gib_run_in_face_slices(flm, weights68):
    """The fp32 kernel addresses its operands through a 2 GiB buffer window (offsets from 0x80000000 mean zero padding):
    at 512 faces f1 is exactly 2 GiB, so enc2 runs as two launches over whole faces.  Faces are independent rows of
    every GEMM: the landmarks must be the bits of the same faces run 64 at a time."""
    from flm_amd.networks import LANDMARKS_MODELS
    n = 512
    crops = np.random.default_rng(7).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256)
    model.load_weights(weights68)
    xd = torch.from_numpy(crops).cuda()
    lm = model.forward_device(xd, "landmarks", n_points=4).cpu().numpy()
    model._ws.clear()
    for lo in (0, 448):     # the first batch of 64 and the one that holds the second slice's only face
        part = model.forward_device(xd[lo:lo + 64].contiguous(), "landmarks", n_points=4).cpu().numpy()
        assert np.array_equal(part, lm[lo:lo + 64]), lo
    assert np.isfinite(lm).all() and (lm >= 0).all()


def test_forward_generic_classes(flm):
    from flm_amd.weights import synth_fcn8_weights
    for c in (5, 21):
        _run_case(flm, synth_fcn8_weights(c, seed=10 + c), n=3, h=64, w=64, c=c, seed=c, u8=True)


def test_predict_numpy_contract(flm, weights68):
    from flm_amd.networks import LANDMARKS_MODELS
    from oracle import fcn_ref
    rng = np.random.default_rng(5)
    model = LANDMARKS_MODELS["default"](68, input_height=32, input_width=64)
    assert (model.output_height, model.output_width, model.n_classes) == (40, 72, 68)
    assert (model.input_height, model.input_width, model.model_name) == (32, 64, "fcn_8")
    assert model.load_weights(weights68) is None
    x = (rng.standard_normal((5, 32, 64, 3)) * 40).astype(np.float32)
    pr = model.predict(x, batch_size=2)
    assert pr.shape == (5, 40 * 72, 68) and pr.dtype == np.float32
    assert np.abs(pr - fcn_ref.fcn8_predict_ref(x, weights68)).max() <= 1e-5


def test_bf16_forward_against_fp32_oracle(flm, weights68):
    """BASELINE configs[2] arithmetic (bf16 operands, fp32 accumulate) at the BASELINE input size.

    Not gated at 1e-4: bf16 has 8 significant bits, so every layer's inputs and weights carry a 2^-9
    relative rounding.  What is checked: the network is the same network (intermediates within a few
    bf16 roundings of the fp32 oracle), the probabilities stay a distribution, and the all-pixel
    landmark centroid stays within half a pixel; the measured NME is printed for the record."""
    from flm_amd import prediction
    from flm_amd.networks import LANDMARKS_MODELS
    from oracle import decode_ref, fcn_ref
    rng = np.random.default_rng(21)
    n, h, w, c = 2, 256, 256, 68
    model = LANDMARKS_MODELS["fcn_8"](c, input_height=h, input_width=w, dtype="bf16")
    model.load_weights(weights68)
    img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    xd = torch.from_numpy(img).cuda()
    x_ref = np.stack([fcn_ref.get_image_array_ref(im) for im in img])
    logits_ref, inter = fcn_ref.fcn8_logits_ref(x_ref, weights68, torch.float32, return_intermediates=True)
    probs = model.forward_device(xd, "probs").cpu().numpy()
    for name, tol in (("f1", 1e-2), ("f2", 2e-2), ("f3", 2e-2), ("f4", 3e-2), ("f5", 3e-2), ("fc6", 4e-2),
                      ("fc7", 4e-2), ("fuse4", 5e-2), ("seg_feats", 5e-2)):
        got = model.intermediate(name, n, "probs").cpu().numpy()[..., : inter[name].shape[-1]]
        assert _rel(got, inter[name]) < tol, (name, _rel(got, inter[name]))
    probs_ref = fcn_ref.fcn8_predict_ref(x_ref, weights68)
    assert np.abs(probs.sum(-1) - 1).max() < 1e-5
    assert np.abs(probs - probs_ref).max() < 0.05
    lm = prediction.predict(xd, model, n_points=0).cpu().numpy()
    with np.errstate(all="ignore"):
        exp = decode_ref.transfer_target_ref(probs_ref.reshape(n, 264, 264, c), 0, 0).reshape(n, c, 2)
    err = np.linalg.norm(lm - exp, axis=-1)
    print("bf16: probs max-abs err %.3g, all-pixel landmark max err %.3g px, NME %.3g" %
          (np.abs(probs - probs_ref).max(), err.max(), err.mean() / 256))
    assert err.max() < 0.5
    cm = model.forward_device(xd, "classmap").cpu().numpy()
    assert (cm == probs_ref.reshape(n, 264, 264, c).argmax(-1)).mean() > 0.9


def test_bf16_256_row_tiles_equal_128_row_tiles(flm, weights68):
    """The 256-row bf16 implicit-GEMM tiles (flm_igemm_bf16.hip) consume k in the same order with the same MFMA
    as the 128x128 kernel, so every intermediate and the probabilities must be bit-identical between the two
    (ragged M: 3 faces of 96x160 leave partial tiles in every layer; fc6 runs position-major with skipped taps)."""
    from flm_amd import _lib
    from flm_amd.networks import LANDMARKS_MODELS
    lib = _lib.load()
    rng = np.random.default_rng(33)
    for (n, h, w) in ((3, 96, 160), (5, 256, 256)):
        model = LANDMARKS_MODELS["fcn_8"](68, input_height=h, input_width=w, dtype="bf16")
        model.load_weights(weights68)
        xd = torch.from_numpy(rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
        outs = {}
        try:
            # 0: 128x128 tiles; (2, 0): 256-row tiles staged through registers; (2, 1): filled by LDS-DMA, 32x32x16 MFMAs;
            # (2, 2): LDS-DMA with 16x16x32 MFMAs (the default; the shape sums a k-run of 32 in one instruction)
            for mode, dma in ((0, 0), (2, 0), (2, 1), (2, 2)):
                _lib.check(lib.flm_set_tuning(b"bf16_big_tiles", mode), "set_tuning")
                _lib.check(lib.flm_set_tuning(b"bf16_lds_dma", 1 if dma else 0), "set_tuning")
                _lib.check(lib.flm_set_tuning(b"bf16_mfma16", 1 if dma == 2 else 0), "set_tuning")
                probs = model.forward_device(xd, "probs").cpu().numpy()
                inter = {k: model.intermediate(k, n, "probs").cpu().numpy() for k in ("f2", "f3", "f4", "f5", "fc6", "fc7")}
                outs[(mode, dma)] = (probs, inter)
        finally:
            _lib.check(lib.flm_set_tuning(b"bf16_big_tiles", 1), "set_tuning")
            _lib.check(lib.flm_set_tuning(b"bf16_lds_dma", 1), "set_tuning")
            _lib.check(lib.flm_set_tuning(b"bf16_mfma16", 1), "set_tuning")
        for key in ((2, 0), (2, 1), (2, 2)):
            for k in outs[(0, 0)][1]:
                assert np.array_equal(outs[(0, 0)][1][k], outs[key][1][k]), (k, key, n, h, w)
            assert np.array_equal(outs[(0, 0)][0], outs[key][0]), key


def test_bf16_score_kernel_equals_implicit_gemm(flm, weights68):
    """flm_score1x1.hip (1x1 classifiers on 256-channel bf16 maps, weights in registers, 32-deep MFMAs) against the
    implicit GEMM (16-deep MFMAs, same k order): fuse4, seg_feats and the probabilities bit for bit; M is ragged
    (3 faces of 96x160: 3 * 12 * 20 = 720 pixels on f3, 45 slices of 16; 180 pixels on f4, the last slice 4 rows)."""
    from flm_amd import _lib
    from flm_amd.networks import LANDMARKS_MODELS
    lib = _lib.load()
    rng = np.random.default_rng(35)
    for (n, h, w) in ((3, 96, 160), (2, 256, 256), (1, 32, 32)):
        model = LANDMARKS_MODELS["fcn_8"](68, input_height=h, input_width=w, dtype="bf16")
        model.load_weights(weights68)
        xd = torch.from_numpy(rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
        outs = {}
        try:
            _lib.check(lib.flm_set_tuning(b"bf16_fused_tail", 0), "set_tuning")   # (score3 as a launch of its own)
            for knob in (0, 1):
                _lib.check(lib.flm_set_tuning(b"bf16_score1x1", knob), "set_tuning")
                probs = model.forward_device(xd, "probs").cpu().numpy()
                inter = {k: model.intermediate(k, n, "probs").cpu().numpy() for k in ("fuse4", "seg_feats")}
                outs[knob] = (probs, inter)
        finally:
            _lib.check(lib.flm_set_tuning(b"bf16_score1x1", 1), "set_tuning")
            _lib.check(lib.flm_set_tuning(b"bf16_fused_tail", 1), "set_tuning")
        for k in outs[0][1]:
            assert np.array_equal(outs[0][1][k], outs[1][1][k]), (k, n, h, w)
        assert np.array_equal(outs[0][0], outs[1][0]), (n, h, w)


def test_bf16_fused_skip_stage_equals_the_two_launches(flm, weights68):
    """flm_tail_bf16.hip: seg_feats = crop(up4(fuse4)) + score3(f3) in one launch against score3 followed by up4 with the
    skip add -- seg_feats, the probabilities and the top-4 landmarks bit for bit; ragged widths (w/16 = 10: one slice of
    10 pixels per row; 2; 32: two slices per row) and the headline shape."""
    from flm_amd import _lib
    from flm_amd.networks import LANDMARKS_MODELS
    lib = _lib.load()
    rng = np.random.default_rng(36)
    for (n, h, w) in ((3, 96, 160), (5, 256, 256), (1, 32, 32), (2, 64, 512)):
        model = LANDMARKS_MODELS["fcn_8"](68, input_height=h, input_width=w, dtype="bf16")
        model.load_weights(weights68)
        xd = torch.from_numpy(rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
        outs = {}
        try:
            for knob in (0, 1):
                _lib.check(lib.flm_set_tuning(b"bf16_fused_tail", knob), "set_tuning")
                probs = model.forward_device(xd, "probs").cpu().numpy()
                seg = model.intermediate("seg_feats", n, "probs").cpu().numpy()
                lm = model.forward_device(xd, "landmarks", n_points=4).cpu().numpy()
                outs[knob] = (probs, seg, lm)
        finally:
            _lib.check(lib.flm_set_tuning(b"bf16_fused_tail", 1), "set_tuning")
        for a, b, what in zip(outs[0], outs[1], ("probs", "seg_feats", "landmarks")):
            assert np.array_equal(a, b), (what, n, h, w)


def test_bf16_halo_conv_equals_implicit_gemm(flm, weights68):
    """flm_conv3_halo.hip (3x3, 64 input channels, halo + weights resident in LDS) consumes k in the implicit GEMM's
    order: bit-identical outputs.  Vanilla enc2 exercises the pooled epilogue, VGG's block1_conv2 / block2_conv1 the
    pooled and the un-pooled one."""
    from flm_amd import _lib
    from flm_amd.networks import LANDMARKS_MODELS
    from flm_amd.weights import synth_vgg_weights
    lib = _lib.load()
    rng = np.random.default_rng(34)
    cases = (("fcn_8", weights68, 3, 96, 160), ("fcn_8", weights68, 2, 256, 256),
             ("fcn_8_vgg", synth_vgg_weights(68, seed=5), 2, 64, 96))
    for name, wts, n, h, w in cases:
        model = LANDMARKS_MODELS[name](68, input_height=h, input_width=w, dtype="bf16")
        model.load_weights(wts)
        xd = torch.from_numpy(rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
        outs = {}
        try:
            # (0, .): implicit GEMM; (2, 0): halo kernel on 32x32x16 MFMAs; (2, 1): on 16x16x32 (the default shape)
            for mode, m16 in ((0, 0), (2, 0), (2, 1)):
                _lib.check(lib.flm_set_tuning(b"bf16_conv3_halo", mode), "set_tuning")
                _lib.check(lib.flm_set_tuning(b"bf16_halo_mfma16", m16), "set_tuning")
                key = (mode, m16)
                outs[key] = model.forward_device(xd, "probs").cpu().numpy()
                if name == "fcn_8":
                    outs[key] = (outs[key], model.intermediate("f2", n, "probs").cpu().numpy())
        finally:
            _lib.check(lib.flm_set_tuning(b"bf16_conv3_halo", 1), "set_tuning")
            _lib.check(lib.flm_set_tuning(b"bf16_halo_mfma16", 1), "set_tuning")
        for key in ((2, 0), (2, 1)):
            if name == "fcn_8":
                assert np.array_equal(outs[(0, 0)][1], outs[key][1]), (name, n, h, w, "f2", key)
                assert np.array_equal(outs[(0, 0)][0], outs[key][0]), (name, n, h, w, key)
            else:
                assert np.array_equal(outs[(0, 0)], outs[key]), (name, n, h, w, key)


def test_fcn32_forward(flm, weights68):
    """fcn_32 (networks/fcn.py:129-150): 64x64 stride-32 transposed conv, output grid H+32."""
    from flm_amd.networks import LANDMARKS_MODELS
    from flm_amd.weights import synth_fcn32_weights
    from oracle import fcn_ref
    rng = np.random.default_rng(32)
    for (n, h, w, c) in ((2, 64, 96, 68), (1, 256, 256, 68), (2, 64, 64, 5)):
        params = synth_fcn32_weights(c, seed=2)
        model = LANDMARKS_MODELS["fcn_32"](c, input_height=h, input_width=w)
        assert (model.output_height, model.output_width, model.model_name) == (h + 32, w + 32, "fcn_32")
        model.load_weights(params)
        img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        x_ref = np.stack([fcn_ref.get_image_array_ref(im) for im in img])
        exp = fcn_ref.fcn32_predict_ref(x_ref, params)
        got = model.forward_device(torch.from_numpy(img).cuda(), "probs").cpu().numpy()
        assert got.shape == exp.shape == (n, (h + 32) * (w + 32), c)
        assert np.abs(got - exp).max() <= 1e-5, (n, h, w, c, np.abs(got - exp).max())


def test_vgg_variants(flm):
    """fcn_8_vgg / fcn_32_vgg (networks/fcn.py:153-164 on networks/vgg16.py:17-81, no pretrained download):
    13 conv3x3+ReLU encoder layers, 512-channel f4/f5, same head and decoders."""
    from flm_amd.networks import LANDMARKS_MODELS
    from flm_amd.weights import synth_vgg_weights
    from oracle import fcn_ref
    rng = np.random.default_rng(33)
    for name, fcn32, (n, h, w, c) in (("fcn_8_vgg", False, (2, 64, 96, 68)), ("fcn_32_vgg", True, (1, 64, 64, 68)),
                                      ("fcn_8_vgg", False, (1, 32, 32, 5))):
        params = synth_vgg_weights(c, seed=4, fcn32=fcn32)
        model = LANDMARKS_MODELS[name](c, input_height=h, input_width=w)
        model.load_weights(params)
        img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        x_ref = np.stack([fcn_ref.get_image_array_ref(im) for im in img])
        ref = fcn_ref.fcn32_predict_ref if fcn32 else fcn_ref.fcn8_predict_ref
        exp = ref(x_ref, params, encoder="vgg")
        got = model.forward_device(torch.from_numpy(img).cuda(), "probs").cpu().numpy()
        assert got.shape == exp.shape
        assert np.abs(got - exp).max() <= 1e-5, (name, np.abs(got - exp).max())


def test_mobilenet_variants(flm):
    """fcn_8_mobilenet / fcn_32_mobilenet (networks/fcn.py:181-192 on networks/mobilenet.py:59-114): stride-2
    conv1, 13 depthwise-separable blocks with BN + ReLU6, 1024-channel f5."""
    from flm_amd.networks import LANDMARKS_MODELS
    from flm_amd.weights import synth_mobilenet_weights
    from oracle import fcn_ref
    rng = np.random.default_rng(34)
    for name, fcn32, (n, h, w, c) in (("fcn_8_mobilenet", False, (2, 64, 96, 68)),
                                      ("fcn_32_mobilenet", True, (1, 64, 64, 68)),
                                      ("fcn_8_mobilenet", False, (1, 224, 224, 68))):
        params = synth_mobilenet_weights(c, seed=5, fcn32=fcn32)
        model = LANDMARKS_MODELS[name](c, input_height=h, input_width=w)
        model.load_weights(params)
        img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        x_ref = np.stack([fcn_ref.get_image_array_ref(im) for im in img])
        ref = fcn_ref.fcn32_predict_ref if fcn32 else fcn_ref.fcn8_predict_ref
        exp = ref(x_ref, params, encoder="mobilenet")
        got = model.forward_device(torch.from_numpy(img).cuda(), "probs").cpu().numpy()
        assert got.shape == exp.shape
        assert np.abs(got - exp).max() <= 1e-5, (name, np.abs(got - exp).max())


def test_mobilenet_bf16_close_to_fp32(flm):
    """bf16 MobileNet encoder: bf16 stem / depthwise kernels, pointwise convs on the bf16 implicit GEMM.  The
    32-channel pointwise conv of block 1 runs on pixel pairs against a block-diagonal filter (flm_pack.hip)."""
    from flm_amd.networks import LANDMARKS_MODELS
    from flm_amd.weights import synth_mobilenet_weights
    rng = np.random.default_rng(37)
    for name, fcn32, (n, h, w) in (("fcn_8_mobilenet", False, (2, 224, 224)), ("fcn_32_mobilenet", True, (1, 64, 96)),
                                   ("fcn_8_mobilenet", False, (3, 96, 160))):
        params = synth_mobilenet_weights(68, seed=5, fcn32=fcn32)
        xd = torch.from_numpy(rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
        out = {}
        for dtype in ("f32", "bf16"):
            model = LANDMARKS_MODELS[name](68, input_height=h, input_width=w, dtype=dtype)
            model.load_weights(params)
            out[dtype] = model.forward_device(xd, "probs").cpu().numpy()
        assert np.isfinite(out["bf16"]).all()
        assert np.abs(out["bf16"].sum(-1) - 1).max() < 1e-5
        d = np.abs(out["bf16"] - out["f32"])
        print(name, (n, h, w), "bf16 vs fp32 probs: max %.3g mean %.3g" % (d.max(), d.mean()))
        assert d.mean() < 2e-3 and d.max() < 0.2, (name, d.max(), d.mean())


def test_resnet50_variants(flm):
    """fcn_8_resnet50 / fcn_32_resnet50 (networks/fcn.py:167-178 on networks/resnet50.py:122-182): 7x7/s2 stem,
    3x3/s2 valid max-pool (odd 63x63 grid at 256x256), 16 bottleneck blocks with strided 1x1 convs and fused
    residual adds, 2048-channel f5."""
    from flm_amd.networks import LANDMARKS_MODELS
    from flm_amd.weights import synth_resnet50_weights
    from oracle import fcn_ref
    rng = np.random.default_rng(35)
    for name, fcn32, (n, h, w, c) in (("fcn_8_resnet50", False, (2, 64, 96, 68)),
                                      ("fcn_32_resnet50", True, (1, 64, 64, 68)),
                                      ("fcn_8_resnet50", False, (1, 256, 256, 68))):
        params = synth_resnet50_weights(c, seed=6, fcn32=fcn32)
        model = LANDMARKS_MODELS[name](c, input_height=h, input_width=w)
        model.load_weights(params)
        img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        x_ref = np.stack([fcn_ref.get_image_array_ref(im) for im in img])
        ref = fcn_ref.fcn32_predict_ref if fcn32 else fcn_ref.fcn8_predict_ref
        exp = ref(x_ref, params, encoder="resnet50")
        got = model.forward_device(torch.from_numpy(img).cuda(), "probs").cpu().numpy()
        assert got.shape == exp.shape
        assert np.abs(got - exp).max() <= 1e-5, (name, np.abs(got - exp).max())


def test_resnet50_bf16_close_to_fp32(flm):
    """bf16 operands / activations through the 53-conv ResNet50 encoder (strided 1x1 convs and residual adds on the
    bf16 implicit GEMM, bf16 stem and max-pool): the probabilities stay a distribution and close to the exact-fp32
    path; not gated at the fp32 bar (8 significant bits per operand, 50 layers deep)."""
    from flm_amd.networks import LANDMARKS_MODELS
    from flm_amd.weights import synth_resnet50_weights
    rng = np.random.default_rng(36)
    for name, fcn32, (n, h, w) in (("fcn_8_resnet50", False, (2, 256, 256)), ("fcn_32_resnet50", True, (1, 64, 96))):
        params = synth_resnet50_weights(68, seed=6, fcn32=fcn32)
        xd = torch.from_numpy(rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
        out = {}
        for dtype in ("f32", "bf16"):
            model = LANDMARKS_MODELS[name](68, input_height=h, input_width=w, dtype=dtype)
            model.load_weights(params)
            out[dtype] = model.forward_device(xd, "probs").cpu().numpy()
        assert np.isfinite(out["bf16"]).all()
        assert np.abs(out["bf16"].sum(-1) - 1).max() < 1e-5
        d = np.abs(out["bf16"] - out["f32"])
        print(name, "bf16 vs fp32 probs: max %.3g mean %.3g" % (d.max(), d.mean()))
        assert d.mean() < 2e-3 and d.max() < 0.2, (name, d.max(), d.mean())
