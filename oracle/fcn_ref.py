"""CPU restatement of the reference's FCN-8 landmark forward.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference's arithmetic lives in TensorFlow/Keras (absent
here, `requirements.txt` empty, nothing pinned) and the reference ships no tests
or golden outputs.  Every function cites the reference lines it restates; Keras
layer semantics are the documented library defaults:

  Conv2D            cross-correlation, kernel HWIO (kh,kw,in,out), use_bias=True
  ZeroPadding2D(1)  symmetric zero pad, applied AFTER preprocessing
  BatchNormalization  last axis, epsilon=1e-3, inference uses moving statistics:
                      gamma*(x-mean)/sqrt(var+eps)+beta
  MaxPooling2D(2,2) stride 2, 'valid' (floor)
  Dropout           identity at inference
  Conv2DTranspose   kernel (kh,kw,out,in), 'valid', use_bias=False,
                    out[s*i+a, s*j+b, o] += x[i,j,c]*w[a,b,o,c]; size (i-1)*s+k
  Cropping2D(((t,b),(l,r)))  removes rows/cols from the named sides
  softmax           last axis, max-subtracted

Weights are passed as a dict of numpy arrays in Keras layouts:
  enc{1..5}/{kernel,bias,gamma,beta,moving_mean,moving_variance}
  fc6|fc7|score5|score4|score3 / {kernel,bias}
  up5|up4|up3 / kernel
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # keras.layers.BatchNormalization default epsilon
MEANS_BGR = (103.939, 116.779, 123.68)  # data/generator.py:56


def get_image_array_ref(img_bgr_u8: np.ndarray) -> np.ndarray:
    """`get_image_array(..., imgNorm="sub_mean", ordering="channels_last")`.

    Follows data/generator.py:52-61 for an input already at (H, W): float32
    cast, per-channel mean subtraction on BGR, then channel reversal (-> RGB).
    The cv2.resize at :53 is the identity when the crop already has the model's
    input size, which is the only case this oracle covers.
    """
    img = img_bgr_u8.astype(np.float32)
    img = np.atleast_3d(img).copy()
    for i in range(min(img.shape[2], 3)):
        img[:, :, i] -= np.float32(MEANS_BGR[i])
    return np.ascontiguousarray(img[:, :, ::-1])


def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def _conv(x, w_hwio, b, pad, dtype):
    # Keras HWIO -> torch OIHW; cross-correlation in both.
    w = _t(w_hwio, dtype).permute(3, 2, 0, 1).contiguous()
    return F.conv2d(x, w, None if b is None else _t(b, dtype), padding=pad)


def _bn(x, p, name, dtype):
    g = _t(p[name + "/gamma"], dtype)[None, :, None, None]
    b = _t(p[name + "/beta"], dtype)[None, :, None, None]
    m = _t(p[name + "/moving_mean"], dtype)[None, :, None, None]
    v = _t(p[name + "/moving_variance"], dtype)[None, :, None, None]
    return g * (x - m) / torch.sqrt(v + BN_EPS) + b


def _convT(x, w_hwoi, stride, dtype):
    # Keras (kh,kw,out,in) -> torch (in,out,kh,kw); both are scatter-add.
    w = _t(w_hwoi, dtype).permute(3, 2, 0, 1).contiguous()
    return F.conv_transpose2d(x, w, None, stride=stride)


def vanilla_encoder_ref(x_nchw, p, dtype):
    """networks/fcn.py:10-51: 5 x (ZeroPadding2D(1), Conv2D 3x3 valid, BN, ReLU, MaxPool 2x2)."""
    levels = []
    x = x_nchw
    for i in range(1, 6):
        n = "enc%d" % i
        x = _conv(x, p[n + "/kernel"], p[n + "/bias"], 1, dtype)  # fcn.py:25-27 / 33-35 / 42-44
        x = _bn(x, p, n, dtype)                                   # fcn.py:28 / 36 / 45
        x = torch.relu(x)                                         # fcn.py:29 / 37 / 46
        x = F.max_pool2d(x, 2, 2)                                 # fcn.py:30 / 38 / 47-48
        levels.append(x)
    return levels


def vgg_encoder_ref(x_nchw, p, dtype):
    """networks/vgg16.py:27-72 (pretrained=None): blocks of Conv2D(3x3, 'same', relu) closed by MaxPooling2D(2,2)."""
    levels = []
    x = x_nchw
    for b, k in ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3)):
        for c in range(1, k + 1):
            n = "block%d_conv%d" % (b, c)
            x = torch.relu(_conv(x, p[n + "/kernel"], p[n + "/bias"], 1, dtype))
        x = F.max_pool2d(x, 2, 2)
        levels.append(x)
    return levels


def mobilenet_encoder_ref(x_nchw, p, dtype):
    """networks/mobilenet.py:59-102 (alpha 1, depth_multiplier 1, pretrained=None): conv1 = ZeroPadding2D(1) +
    Conv2D(32, 3x3, stride 2, valid, no bias) + BN + ReLU6 (:16-29); 13 blocks of ZeroPadding2D(1) +
    DepthwiseConv2D(3x3, stride s, valid, no bias) + BN + ReLU6 + Conv2D(1x1, no bias) + BN + ReLU6 (:32-56);
    strides 2 at blocks 2, 4, 6, 12; levels after blocks 1, 3, 5, 11, 13."""
    def relu6(t):
        return torch.clamp(t, 0.0, 6.0)

    w = _t(p["conv1/kernel"], dtype).permute(3, 2, 0, 1).contiguous()
    x = relu6(_bn(F.conv2d(x_nchw, w, None, stride=2, padding=1), p, "conv1_bn", dtype))
    levels = []
    for i in range(1, 14):
        s = 2 if i in (2, 4, 6, 12) else 1
        dw = _t(p["conv_dw_%d/depthwise_kernel" % i], dtype).permute(2, 3, 0, 1).contiguous()  # (C,1,3,3)
        x = relu6(_bn(F.conv2d(x, dw, None, stride=s, padding=1, groups=dw.shape[0]), p, "conv_dw_%d_bn" % i, dtype))
        pw = _t(p["conv_pw_%d/kernel" % i], dtype).permute(3, 2, 0, 1).contiguous()
        x = relu6(_bn(F.conv2d(x, pw, None), p, "conv_pw_%d_bn" % i, dtype))
        if i in (1, 3, 5, 11, 13):
            levels.append(x)
    return levels


def resnet50_encoder_ref(x_nchw, p, dtype):
    """networks/resnet50.py:145-170 (pretrained=None): ZeroPadding2D(3) + Conv2D(64, 7x7, s2) + BN + ReLU +
    MaxPooling2D(3x3, s2, valid); stages 2..5 of bottleneck blocks (conv_block :81-118 with a 1x1 shortcut conv,
    stride on branch2a and the shortcut; identity_block :33-78); every Conv2D has a bias.  Returns
    [f1-like, f2-like, f3, f4, f5]; only f3..f5 feed the FCN decoders (f1 is pre-BN and f2 one-side padded in
    the reference, and unused)."""
    def cbn(x, conv, bn, k, stride, relu, res=None):
        w = _t(p[conv + "/kernel"], dtype).permute(3, 2, 0, 1).contiguous()
        y = F.conv2d(x, w, _t(p[conv + "/bias"], dtype), stride=stride, padding=k // 2)
        y = _bn(y, p, bn, dtype)
        if res is not None:
            y = y + res
        return torch.relu(y) if relu else y

    x = cbn(x_nchw, "conv1", "bn_conv1", 7, 2, True)
    f1 = x
    x = F.max_pool2d(x, 3, 2)
    levels = [f1]
    for stage, blocks in ((2, "abc"), (3, "abcd"), (4, "abcdef"), (5, "abc")):
        for b in blocks:
            base, bn = "res%d%s_branch" % (stage, b), "bn%d%s_branch" % (stage, b)
            s = 2 if (b == "a" and stage > 2) else 1
            shortcut = cbn(x, base + "1", bn + "1", 1, s, False) if b == "a" else x
            y = cbn(x, base + "2a", bn + "2a", 1, s, True)
            y = cbn(y, base + "2b", bn + "2b", 3, 1, True)
            x = cbn(y, base + "2c", bn + "2c", 1, 1, True, res=shortcut)
        levels.append(x)
    return levels


def _encoder(name):
    return {"vgg": vgg_encoder_ref, "mobilenet": mobilenet_encoder_ref,
            "resnet50": resnet50_encoder_ref}.get(name, vanilla_encoder_ref)


def crop_ref(o1, o2):
    """networks/fcn.py:55-86.  The larger map loses its RIGHT columns and BOTTOM rows
    (Cropping2D(((0,0),(0,cx))) then Cropping2D(((0,cy),(0,0)))): the top-left window stays."""
    h1, w1 = o1.shape[2], o1.shape[3]
    h2, w2 = o2.shape[2], o2.shape[3]
    cx, cy = abs(w1 - w2), abs(h2 - h1)
    if w1 > w2:
        o1 = o1[:, :, :, : w1 - cx]
    else:
        o2 = o2[:, :, :, : w2 - cx]
    if h1 > h2:
        o1 = o1[:, :, : h1 - cy, :]
    else:
        o2 = o2[:, :, : h2 - cy, :]
    return o1, o2


def fcn8_logits_ref(x_nhwc: np.ndarray, p: dict, dtype=torch.float32, return_intermediates=False, encoder="vanilla"):
    """networks/fcn.py:89-122 up to (not including) the softmax.  x: [N,H,W,3] preprocessed.
    encoder="vgg" is fcn_8_vgg (fcn.py:153-157)."""
    x = _t(x_nhwc, dtype).permute(0, 3, 1, 2).contiguous()
    f1, f2, f3, f4, f5 = _encoder(encoder)(x, p, dtype)
    o = torch.relu(_conv(f5, p["fc6/kernel"], p["fc6/bias"], 3, dtype))   # fcn.py:98 7x7 'same'; :99 dropout = id
    fc6 = o
    o = torch.relu(_conv(o, p["fc7/kernel"], p["fc7/bias"], 0, dtype))    # fcn.py:100-101
    fc7 = o
    o = _conv(o, p["score5/kernel"], p["score5/bias"], 0, dtype)          # fcn.py:103
    o = _convT(o, p["up5/kernel"], 2, dtype)                              # fcn.py:104-105
    o2 = _conv(f4, p["score4/kernel"], p["score4/bias"], 0, dtype)        # fcn.py:107-108
    o, o2 = crop_ref(o, o2)                                               # fcn.py:110
    o = o + o2                                                            # fcn.py:112
    fuse4 = o
    o = _convT(o, p["up4/kernel"], 2, dtype)                              # fcn.py:114-115
    o2 = _conv(f3, p["score3/kernel"], p["score3/bias"], 0, dtype)        # fcn.py:116-117
    o2, o = crop_ref(o2, o)                                               # fcn.py:118
    o = o2 + o                                                            # fcn.py:119 "seg_feats"
    seg = o
    o = _convT(o, p["up3/kernel"], 8, dtype)                              # fcn.py:121-122
    logits = o.permute(0, 2, 3, 1).contiguous()                           # NHWC
    if return_intermediates:
        nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().numpy()
        return logits.numpy(), dict(f1=nhwc(f1), f2=nhwc(f2), f3=nhwc(f3), f4=nhwc(f4), f5=nhwc(f5),
                                    fc6=nhwc(fc6), fc7=nhwc(fc7), fuse4=nhwc(fuse4), seg_feats=nhwc(seg))
    return logits.numpy()


def fcn32_logits_ref(x_nhwc: np.ndarray, p: dict, dtype=torch.float32, encoder="vanilla") -> np.ndarray:
    """networks/fcn.py:129-146 up to the softmax: encoder, fc6, fc7, 1x1 classifier ("seg_feats", :143-144),
    Conv2DTranspose(C, 64x64, stride 32, no bias) (:145-146).  Output grid (H/32 - 1)*32 + 64 = H + 32."""
    x = _t(x_nhwc, dtype).permute(0, 3, 1, 2).contiguous()
    f5 = _encoder(encoder)(x, p, dtype)[4]
    o = torch.relu(_conv(f5, p["fc6/kernel"], p["fc6/bias"], 3, dtype))   # fcn.py:138
    o = torch.relu(_conv(o, p["fc7/kernel"], p["fc7/bias"], 0, dtype))    # fcn.py:140
    o = _conv(o, p["score5/kernel"], p["score5/bias"], 0, dtype)          # fcn.py:143-144
    o = _convT(o, p["up32/kernel"], 32, dtype)                            # fcn.py:145-146
    return o.permute(0, 2, 3, 1).contiguous().numpy()


def fcn32_predict_ref(x_nhwc: np.ndarray, p: dict, dtype=torch.float32, encoder="vanilla") -> np.ndarray:
    """fcn_32 + get_segmentation_model (networks/utils.py:22-31): [N, H'*W', C] probabilities."""
    logits = torch.from_numpy(fcn32_logits_ref(x_nhwc, p, dtype, encoder))
    n, h, w, c = logits.shape
    return torch.softmax(logits.reshape(n, h * w, c), dim=-1).numpy()


def fcn8_predict_ref(x_nhwc: np.ndarray, p: dict, dtype=torch.float32, encoder="vanilla") -> np.ndarray:
    """`model.predict` of the model built by fcn_8 + get_segmentation_model
    (networks/utils.py:22-31): Reshape((H'*W', C)) then softmax over the last axis.
    Returns [N, H'*W', C]."""
    logits = torch.from_numpy(fcn8_logits_ref(x_nhwc, p, dtype, encoder=encoder))
    n, h, w, c = logits.shape
    pr = torch.softmax(logits.reshape(n, h * w, c), dim=-1)
    return pr.numpy()


def output_hw(input_h: int, input_w: int):
    """Output grid of fcn_8 for an (input_h, input_w) input (multiples of 32):
    enc -> /32; up5 (i-1)*2+4 cropped to /16; up4 cropped to /8; up3 (i-1)*8+16 = H+8."""
    return input_h + 8, input_w + 8


def class_map_ref(pr: np.ndarray, out_h: int, out_w: int, n_classes: int) -> np.ndarray:
    """prediction.py:209: `pr.reshape((H', W', C)).argmax(axis=2)` (first maximum wins)."""
    return pr.reshape((out_h, out_w, n_classes)).argmax(axis=2)


def prediction_ref(img_bgr_u8: np.ndarray, p: dict, n_classes: int):
    """`_prediction` up to the class map (prediction.py:207-209) for one crop
    whose size already equals the model input size."""
    x = get_image_array_ref(img_bgr_u8)
    pr = fcn8_predict_ref(np.array([x]), p)[0]
    oh, ow = output_hw(*img_bgr_u8.shape[:2])
    return class_map_ref(pr, oh, ow, n_classes), pr
