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
