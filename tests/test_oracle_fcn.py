"""Cross-check oracle/fcn_ref.py (torch ops) against an independent naive
restatement of the same Keras semantics written with explicit window loops.

The reference holds no golden outputs for the FCN forward (it delegates to
TensorFlow, absent here): parity of the forward is UNPINNED against the
reference itself; this test guards the oracle against misreadings of
networks/fcn.py:10-126 and networks/utils.py:22-31 by computing everything twice
with unrelated code.
"""
import numpy as np
import torch

from oracle import fcn_ref


def tiny_params(n_classes=5, seed=0, enc=(4, 6, 8, 8, 8), fc=16):
    rng = np.random.default_rng(seed)
    p = {}
    cin = 3
    for i, f in enumerate(enc, 1):
        n = "enc%d" % i
        p[n + "/kernel"] = rng.standard_normal((3, 3, cin, f)).astype(np.float32) * 0.3
        p[n + "/bias"] = rng.standard_normal(f).astype(np.float32) * 0.1
        p[n + "/gamma"] = rng.uniform(0.5, 1.5, f).astype(np.float32)
        p[n + "/gamma"][0] *= -1  # a negative BN scale: BN must precede ReLU/pool
        p[n + "/beta"] = rng.standard_normal(f).astype(np.float32) * 0.1
        p[n + "/moving_mean"] = rng.standard_normal(f).astype(np.float32) * 0.1
        p[n + "/moving_variance"] = rng.uniform(0.5, 1.5, f).astype(np.float32)
        cin = f
    p["fc6/kernel"] = rng.standard_normal((7, 7, cin, fc)).astype(np.float32) * 0.1
    p["fc6/bias"] = rng.standard_normal(fc).astype(np.float32) * 0.1
    p["fc7/kernel"] = rng.standard_normal((1, 1, fc, fc)).astype(np.float32) * 0.3
    p["fc7/bias"] = rng.standard_normal(fc).astype(np.float32) * 0.1
    p["score5/kernel"] = rng.standard_normal((1, 1, fc, n_classes)).astype(np.float32) * 0.3
    p["score5/bias"] = rng.standard_normal(n_classes).astype(np.float32) * 0.1
    p["score4/kernel"] = rng.standard_normal((1, 1, enc[3], n_classes)).astype(np.float32) * 0.3
    p["score4/bias"] = rng.standard_normal(n_classes).astype(np.float32) * 0.1
    p["score3/kernel"] = rng.standard_normal((1, 1, enc[2], n_classes)).astype(np.float32) * 0.3
    p["score3/bias"] = rng.standard_normal(n_classes).astype(np.float32) * 0.1
    p["up5/kernel"] = rng.standard_normal((4, 4, n_classes, n_classes)).astype(np.float32) * 0.3
    p["up4/kernel"] = rng.standard_normal((4, 4, n_classes, n_classes)).astype(np.float32) * 0.3
    p["up3/kernel"] = rng.standard_normal((16, 16, n_classes, n_classes)).astype(np.float32) * 0.1
    return p


# ---- naive restatement (float64, explicit windows) ---------------------------------------

def n_conv(x, w, b, pad):
    kh, kw, ci, co = w.shape
    xp = np.pad(x, ((pad, pad), (pad, pad), (0, 0)))
    H, W = xp.shape[0] - kh + 1, xp.shape[1] - kw + 1
    out = np.zeros((H, W, co))
    for ky in range(kh):
        for kx in range(kw):
            out += xp[ky:ky + H, kx:kx + W, :] @ w[ky, kx].astype(np.float64)
    return out + b


def n_bn(x, p, n):
    return p[n + "/gamma"] * (x - p[n + "/moving_mean"]) / np.sqrt(p[n + "/moving_variance"].astype(np.float64) + 1e-3) + p[n + "/beta"]


def n_pool(x):
    H, W, C = x.shape
    return x[:H // 2 * 2, :W // 2 * 2].reshape(H // 2, 2, W // 2, 2, C).max(axis=(1, 3))


def n_convT(x, w, s):
    kh, kw, co, ci = w.shape
    H, W, _ = x.shape
    out = np.zeros(((H - 1) * s + kh, (W - 1) * s + kw, co))
    for i in range(H):
        for j in range(W):
            out[s * i:s * i + kh, s * j:s * j + kw, :] += np.einsum("c,aboc->abo", x[i, j], w.astype(np.float64))
    return out


def naive_fcn8(x, p):
    levels = []
    for i in range(1, 6):
        n = "enc%d" % i
        x = n_pool(np.maximum(n_bn(n_conv(x, p[n + "/kernel"], p[n + "/bias"], 1), p, n), 0))
        levels.append(x)
    f1, f2, f3, f4, f5 = levels
    o = np.maximum(n_conv(f5, p["fc6/kernel"], p["fc6/bias"], 3), 0)
    o = np.maximum(n_conv(o, p["fc7/kernel"], p["fc7/bias"], 0), 0)
    o = n_conv(o, p["score5/kernel"], p["score5/bias"], 0)
    o = n_convT(o, p["up5/kernel"], 2)
    o2 = n_conv(f4, p["score4/kernel"], p["score4/bias"], 0)
    o = o[:o2.shape[0], :o2.shape[1]] + o2          # crop keeps the top-left window
    o = n_convT(o, p["up4/kernel"], 2)
    o2 = n_conv(f3, p["score3/kernel"], p["score3/bias"], 0)
    o = o[:o2.shape[0], :o2.shape[1]] + o2
    o = n_convT(o, p["up3/kernel"], 8)
    z = o - o.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return o, (e / e.sum(axis=-1, keepdims=True)).reshape(-1, o.shape[-1])


def test_shapes_for_256():
    # fcn.py:121-124: (32-1)*8+16 = 264, no final crop
    assert fcn_ref.output_hw(256, 256) == (264, 264)
    assert fcn_ref.output_hw(416, 608) == (424, 616)


def test_oracle_matches_naive_restatement():
    p = tiny_params()
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 64, 96, 3)).astype(np.float32)
    logits64 = fcn_ref.fcn8_logits_ref(x, p, torch.float64)
    probs64 = fcn_ref.fcn8_predict_ref(x, p, torch.float64)
    assert logits64.shape == (2, 72, 104, 5)
    for n in range(2):
        lo, pr = naive_fcn8(x[n].astype(np.float64), p)
        np.testing.assert_allclose(logits64[n], lo, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(probs64[n], pr, rtol=1e-9, atol=1e-12)
    # fp32 oracle stays within fp32 rounding of the fp64 one
    probs32 = fcn_ref.fcn8_predict_ref(x, p, torch.float32)
    assert np.abs(probs32 - probs64).max() < 1e-5


def test_crop_keeps_top_left():
    a = torch.arange(2 * 1 * 5 * 6, dtype=torch.float32).reshape(2, 1, 5, 6)
    b = torch.zeros(2, 1, 4, 4)
    a2, b2 = fcn_ref.crop_ref(a, b)
    assert a2.shape == (2, 1, 4, 4) and b2.shape == (2, 1, 4, 4)
    assert torch.equal(a2, a[:, :, :4, :4])


def test_preprocess_and_classmap():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
    x = fcn_ref.get_image_array_ref(img)
    assert x.dtype == np.float32 and x.shape == (8, 8, 3)
    # channel 0 of the output is R = input channel 2 minus 123.68 (generator.py:56-61)
    assert x[3, 4, 0] == np.float32(img[3, 4, 2]) - np.float32(123.68)
    assert x[3, 4, 2] == np.float32(img[3, 4, 0]) - np.float32(103.939)
    pr = np.zeros((6, 3), np.float32)
    pr[:, 1] = 1
    pr[2] = [0.5, 0.5, 0]  # tie -> first maximum
    cm = fcn_ref.class_map_ref(pr, 2, 3, 3)
    assert cm.dtype == np.int64 and cm[0, 2] == 0 and cm[1, 1] == 1


def test_fcn32_oracle_matches_naive():
    p = tiny_params()
    rng = np.random.default_rng(3)
    p["up32/kernel"] = rng.standard_normal((64, 64, 5, 5)).astype(np.float32) * 0.1
    x = rng.standard_normal((1, 64, 96, 3)).astype(np.float32)
    got = fcn_ref.fcn32_logits_ref(x, p, torch.float64)
    assert got.shape == (1, 96, 128, 5)     # (H/32 - 1)*32 + 64 = H + 32
    xx = x[0].astype(np.float64)
    for i in range(1, 6):
        n = "enc%d" % i
        xx = n_pool(np.maximum(n_bn(n_conv(xx, p[n + "/kernel"], p[n + "/bias"], 1), p, n), 0))
    o = np.maximum(n_conv(xx, p["fc6/kernel"], p["fc6/bias"], 3), 0)
    o = np.maximum(n_conv(o, p["fc7/kernel"], p["fc7/bias"], 0), 0)
    o = n_conv(o, p["score5/kernel"], p["score5/bias"], 0)
    o = n_convT(o, p["up32/kernel"], 32)
    np.testing.assert_allclose(got[0], o, rtol=1e-9, atol=1e-9)
