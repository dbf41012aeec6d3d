"""Weight container for the FCN-8 landmark model.

The reference stores TF-checkpoint files written by Keras `ModelCheckpoint`
(training.py:218-222); those cannot be parsed without TensorFlow and none ship
with the reference.  This build's container is a flat ``.npz`` whose keys are
``<layer>/<tensor>`` with tensors in the KERAS layouts (so a checkpoint exported
from Keras with `layer.get_weights()` drops in unchanged):

  enc{1..5}/kernel [3,3,Cin,F] HWIO   enc{i}/bias [F]
  enc{i}/gamma|beta|moving_mean|moving_variance [F]     (BatchNormalization)
  fc6/kernel [7,7,256,4096]  fc7/kernel [1,1,4096,4096]  + bias
  score5/kernel [1,1,4096,C] score4/kernel [1,1,256,C] score3/kernel [1,1,256,C] + bias
  up5/kernel [4,4,C,C] up4/kernel [4,4,C,C] up3/kernel [16,16,C,C]   (kh,kw,out,in), no bias

No trained weights exist anywhere (reference README.md:4-7 lists the FCN detector
as TODO), so benchmarks and tests use the seeded synthetic set below.
"""
from __future__ import annotations

import numpy as np

ENC_FILTERS = (64, 128, 256, 256, 256)  # networks/fcn.py:13,34,43
FC_WIDTH = 4096                          # networks/fcn.py:98,100


VGG_BLOCKS = ((1, 2, 64), (2, 2, 128), (3, 3, 256), (4, 3, 512), (5, 3, 512))  # networks/vgg16.py:27-72


def vgg_param_shapes(n_classes: int, channels: int = 3, fcn32: bool = False) -> dict:
    """fcn_8_vgg / fcn_32_vgg: 13 conv3x3 (Keras names block{b}_conv{c}), head on 512-channel f5/f4, 256-channel f3."""
    shapes = {}
    cin = channels
    for b, k, f in VGG_BLOCKS:
        for c in range(1, k + 1):
            shapes["block%d_conv%d/kernel" % (b, c)] = (3, 3, cin, f)
            shapes["block%d_conv%d/bias" % (b, c)] = (f,)
            cin = f
    shapes["fc6/kernel"] = (7, 7, 512, FC_WIDTH)
    shapes["fc6/bias"] = (FC_WIDTH,)
    shapes["fc7/kernel"] = (1, 1, FC_WIDTH, FC_WIDTH)
    shapes["fc7/bias"] = (FC_WIDTH,)
    shapes["score5/kernel"] = (1, 1, FC_WIDTH, n_classes)
    shapes["score5/bias"] = (n_classes,)
    if fcn32:
        shapes["up32/kernel"] = (64, 64, n_classes, n_classes)
        return shapes
    shapes["score4/kernel"] = (1, 1, 512, n_classes)
    shapes["score4/bias"] = (n_classes,)
    shapes["score3/kernel"] = (1, 1, 256, n_classes)
    shapes["score3/bias"] = (n_classes,)
    shapes["up5/kernel"] = (4, 4, n_classes, n_classes)
    shapes["up4/kernel"] = (4, 4, n_classes, n_classes)
    shapes["up3/kernel"] = (16, 16, n_classes, n_classes)
    return shapes


def synth_vgg_weights(n_classes: int = 68, seed: int = 2, channels: int = 3, fcn32: bool = False) -> dict:
    """Seeded synthetic parameters for the VGG variants (he_normal kernels, small biases; see synth_fcn8_weights)."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shp in vgg_param_shapes(n_classes, channels, fcn32).items():
        layer, tensor = name.split("/")
        if tensor == "bias":
            p[name] = rng.standard_normal(shp, dtype=np.float32) * np.float32(0.01)
            continue
        if layer.startswith("up"):
            kh, kw, co, ci = shp
            stride = {"up3": 8, "up32": 32}.get(layer, 2)
            std = np.sqrt(1.0 / ((kh // stride) * (kw // stride) * ci))
        else:
            kh, kw, ci, co = shp
            std = np.sqrt(2.0 / (kh * kw * ci))
        w = rng.standard_normal(shp, dtype=np.float32) * np.float32(std)
        if layer == "block1_conv1":
            w *= np.float32(1.0 / 64.0)   # mean-subtracted byte inputs
        if layer.startswith("score"):
            w *= np.float32(0.125)
        p[name] = w
    return p


MOBILENET_PW = (64, 128, 128, 256, 256, 512, 512, 512, 512, 512, 512, 1024, 1024)  # networks/mobilenet.py:80-102


def mobilenet_param_shapes(n_classes: int, channels: int = 3, fcn32: bool = False) -> dict:
    """fcn_8_mobilenet / fcn_32_mobilenet: Keras layer names of networks/mobilenet.py; BN under `<layer>_bn`."""
    shapes = {}

    def bn(name, c):
        for t in ("gamma", "beta", "moving_mean", "moving_variance"):
            shapes["%s/%s" % (name, t)] = (c,)

    shapes["conv1/kernel"] = (3, 3, channels, 32)
    bn("conv1_bn", 32)
    cin = 32
    for i, f in enumerate(MOBILENET_PW, 1):
        shapes["conv_dw_%d/depthwise_kernel" % i] = (3, 3, cin, 1)
        bn("conv_dw_%d_bn" % i, cin)
        shapes["conv_pw_%d/kernel" % i] = (1, 1, cin, f)
        bn("conv_pw_%d_bn" % i, f)
        cin = f
    shapes["fc6/kernel"] = (7, 7, 1024, FC_WIDTH)
    shapes["fc6/bias"] = (FC_WIDTH,)
    shapes["fc7/kernel"] = (1, 1, FC_WIDTH, FC_WIDTH)
    shapes["fc7/bias"] = (FC_WIDTH,)
    shapes["score5/kernel"] = (1, 1, FC_WIDTH, n_classes)
    shapes["score5/bias"] = (n_classes,)
    if fcn32:
        shapes["up32/kernel"] = (64, 64, n_classes, n_classes)
        return shapes
    shapes["score4/kernel"] = (1, 1, 512, n_classes)
    shapes["score4/bias"] = (n_classes,)
    shapes["score3/kernel"] = (1, 1, 256, n_classes)
    shapes["score3/bias"] = (n_classes,)
    shapes["up5/kernel"] = (4, 4, n_classes, n_classes)
    shapes["up4/kernel"] = (4, 4, n_classes, n_classes)
    shapes["up3/kernel"] = (16, 16, n_classes, n_classes)
    return shapes


def synth_mobilenet_weights(n_classes: int = 68, seed: int = 2, channels: int = 3, fcn32: bool = False) -> dict:
    """Seeded synthetic parameters for the MobileNet variants."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shp in mobilenet_param_shapes(n_classes, channels, fcn32).items():
        layer, tensor = name.split("/")
        if tensor == "bias":
            p[name] = rng.standard_normal(shp, dtype=np.float32) * np.float32(0.01)
        elif tensor == "gamma":
            p[name] = rng.uniform(0.8, 1.6, shp).astype(np.float32)
        elif tensor in ("beta", "moving_mean"):
            p[name] = rng.standard_normal(shp, dtype=np.float32) * np.float32(0.1)
        elif tensor == "moving_variance":
            p[name] = rng.uniform(0.5, 1.5, shp).astype(np.float32)
        elif tensor == "depthwise_kernel":
            p[name] = rng.standard_normal(shp, dtype=np.float32) * np.float32(np.sqrt(2.0 / 9.0))
        else:
            if layer.startswith("up"):
                kh, kw, co, ci = shp
                stride = {"up3": 8, "up32": 32}.get(layer, 2)
                std = np.sqrt(1.0 / ((kh // stride) * (kw // stride) * ci))
            else:
                kh, kw, ci, co = shp
                std = np.sqrt(2.0 / (kh * kw * ci))
            w = rng.standard_normal(shp, dtype=np.float32) * np.float32(std)
            if layer == "conv1":
                w *= np.float32(1.0 / 64.0)
            if layer.startswith("score"):
                w *= np.float32(0.125)
            p[name] = w
    return p


RESNET_STAGES = ((2, "abc", 64), (3, "abcd", 128), (4, "abcdef", 256), (5, "abc", 512))  # networks/resnet50.py:154-170


def resnet50_conv_layers():
    """(conv layer name, bn layer name, kernel size, cin, cout) in the order the C library consumes them:
    conv1, then per block [shortcut `branch1` if the block is a conv_block], branch2a, 2b, 2c."""
    layers = [("conv1", "bn_conv1", 7, 3, 64)]
    cin = 64
    for stage, blocks, f in RESNET_STAGES:
        for b in blocks:
            base, bn = "res%d%s_branch" % (stage, b), "bn%d%s_branch" % (stage, b)
            if b == "a":
                layers.append((base + "1", bn + "1", 1, cin, 4 * f))
            layers.append((base + "2a", bn + "2a", 1, cin, f))
            layers.append((base + "2b", bn + "2b", 3, f, f))
            layers.append((base + "2c", bn + "2c", 1, f, 4 * f))
            cin = 4 * f
    return layers


def resnet50_param_shapes(n_classes: int, channels: int = 3, fcn32: bool = False) -> dict:
    """fcn_8_resnet50 / fcn_32_resnet50: Keras layer names of networks/resnet50.py (conv + separate bn layers)."""
    shapes = {}
    for conv, bn, k, cin, cout in resnet50_conv_layers():
        shapes[conv + "/kernel"] = (k, k, cin, cout)
        shapes[conv + "/bias"] = (cout,)
        for t in ("gamma", "beta", "moving_mean", "moving_variance"):
            shapes["%s/%s" % (bn, t)] = (cout,)
    shapes["fc6/kernel"] = (7, 7, 2048, FC_WIDTH)
    shapes["fc6/bias"] = (FC_WIDTH,)
    shapes["fc7/kernel"] = (1, 1, FC_WIDTH, FC_WIDTH)
    shapes["fc7/bias"] = (FC_WIDTH,)
    shapes["score5/kernel"] = (1, 1, FC_WIDTH, n_classes)
    shapes["score5/bias"] = (n_classes,)
    if fcn32:
        shapes["up32/kernel"] = (64, 64, n_classes, n_classes)
        return shapes
    shapes["score4/kernel"] = (1, 1, 1024, n_classes)
    shapes["score4/bias"] = (n_classes,)
    shapes["score3/kernel"] = (1, 1, 512, n_classes)
    shapes["score3/bias"] = (n_classes,)
    shapes["up5/kernel"] = (4, 4, n_classes, n_classes)
    shapes["up4/kernel"] = (4, 4, n_classes, n_classes)
    shapes["up3/kernel"] = (16, 16, n_classes, n_classes)
    return shapes


def synth_resnet50_weights(n_classes: int = 68, seed: int = 2, channels: int = 3, fcn32: bool = False) -> dict:
    """Seeded synthetic parameters for the ResNet50 variants.  The last BN of every block gets a small gamma so
    the residual sums stay O(1) through 16 blocks."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shp in resnet50_param_shapes(n_classes, channels, fcn32).items():
        layer, tensor = name.split("/")
        if tensor == "bias":
            p[name] = rng.standard_normal(shp, dtype=np.float32) * np.float32(0.01)
        elif tensor == "gamma":
            g = rng.uniform(0.8, 1.2, shp).astype(np.float32)
            p[name] = g * np.float32(0.3) if layer.endswith("2c") else g
        elif tensor in ("beta", "moving_mean"):
            p[name] = rng.standard_normal(shp, dtype=np.float32) * np.float32(0.1)
        elif tensor == "moving_variance":
            p[name] = rng.uniform(0.5, 1.5, shp).astype(np.float32)
        else:
            if layer.startswith("up"):
                kh, kw, co, ci = shp
                stride = {"up3": 8, "up32": 32}.get(layer, 2)
                std = np.sqrt(1.0 / ((kh // stride) * (kw // stride) * ci))
            else:
                kh, kw, ci, co = shp
                std = np.sqrt(2.0 / (kh * kw * ci))
            w = rng.standard_normal(shp, dtype=np.float32) * np.float32(std)
            if layer == "conv1":
                w *= np.float32(1.0 / 64.0)
            if layer.startswith("score"):
                w *= np.float32(0.125)
            p[name] = w
    return p


def fcn32_param_shapes(n_classes: int, channels: int = 3) -> dict:
    """fcn_32 (networks/fcn.py:129-150): encoder + fc6 + fc7 + 1x1 classifier + one 64x64/s32 transposed conv."""
    s8 = fcn8_param_shapes(n_classes, channels)
    shapes = {k: v for k, v in s8.items() if k.split("/")[0] not in ("score4", "score3", "up5", "up4", "up3")}
    shapes["up32/kernel"] = (64, 64, n_classes, n_classes)
    return shapes


def fcn8_param_shapes(n_classes: int, channels: int = 3) -> dict:
    shapes = {}
    cin = channels
    for i, f in enumerate(ENC_FILTERS, 1):
        n = "enc%d" % i
        shapes[n + "/kernel"] = (3, 3, cin, f)
        for t in ("bias", "gamma", "beta", "moving_mean", "moving_variance"):
            shapes[n + "/" + t] = (f,)
        cin = f
    shapes["fc6/kernel"] = (7, 7, cin, FC_WIDTH)
    shapes["fc6/bias"] = (FC_WIDTH,)
    shapes["fc7/kernel"] = (1, 1, FC_WIDTH, FC_WIDTH)
    shapes["fc7/bias"] = (FC_WIDTH,)
    shapes["score5/kernel"] = (1, 1, FC_WIDTH, n_classes)
    shapes["score5/bias"] = (n_classes,)
    shapes["score4/kernel"] = (1, 1, ENC_FILTERS[3], n_classes)
    shapes["score4/bias"] = (n_classes,)
    shapes["score3/kernel"] = (1, 1, ENC_FILTERS[2], n_classes)
    shapes["score3/bias"] = (n_classes,)
    shapes["up5/kernel"] = (4, 4, n_classes, n_classes)
    shapes["up4/kernel"] = (4, 4, n_classes, n_classes)
    shapes["up3/kernel"] = (16, 16, n_classes, n_classes)
    return shapes


def synth_fcn32_weights(n_classes: int = 68, seed: int = 2, channels: int = 3) -> dict:
    """Seeded synthetic fcn_32 parameters: the fcn_8 set's shared tensors + a scaled 64x64 kernel."""
    p8 = synth_fcn8_weights(n_classes, seed, channels)
    p = {k: v for k, v in p8.items() if k in fcn32_param_shapes(n_classes, channels)}
    rng = np.random.default_rng(seed + 1000)
    # each output pixel sums 2x2 taps of n_classes inputs
    p["up32/kernel"] = (rng.standard_normal((64, 64, n_classes, n_classes), dtype=np.float32)
                        * np.float32(np.sqrt(1.0 / (4 * n_classes))))
    return p


def synth_fcn8_weights(n_classes: int = 68, seed: int = 2, channels: int = 3) -> dict:
    """Seeded synthetic parameters.

    Conv kernels ~ N(0, 2/fan_in) (he_normal, as networks/fcn.py:103,108,117 ask
    for the score convs); biases ~ N(0, 0.01); BN gamma ~ U(0.5,1.5),
    beta ~ N(0,0.1), mean ~ N(0,0.1), var ~ U(0.5,1.5).  The three score convs and
    the transposed convs are scaled so the logits keep an O(1) spread (an
    unsaturated softmax, so argmax / centroid are not degenerate).
    """
    rng = np.random.default_rng(seed)
    p = {}
    for name, shp in fcn8_param_shapes(n_classes, channels).items():
        layer, tensor = name.split("/")
        if tensor == "kernel":
            if layer.startswith("up"):
                kh, kw, co, ci = shp
                stride = 8 if layer == "up3" else 2
                # each output pixel sums (kh/stride)*(kw/stride) taps of ci inputs
                fan = (kh // stride) * (kw // stride) * ci
                std = np.sqrt(1.0 / fan)
            else:
                kh, kw, ci, co = shp
                std = np.sqrt(2.0 / (kh * kw * ci))
            w = rng.standard_normal(shp, dtype=np.float32) * np.float32(std)
            if layer == "enc1":
                # inputs are mean-subtracted bytes (|x| up to ~130): keep enc1 O(1)
                w *= np.float32(1.0 / 64.0)
            if layer.startswith("score"):
                # keep the logits' spread O(1): an unsaturated softmax
                w *= np.float32(0.125)
            p[name] = w
        elif tensor == "bias":
            p[name] = (rng.standard_normal(shp, dtype=np.float32) * np.float32(0.01))
        elif tensor == "gamma":
            p[name] = rng.uniform(0.5, 1.5, shp).astype(np.float32)
        elif tensor == "beta":
            p[name] = (rng.standard_normal(shp, dtype=np.float32) * np.float32(0.1))
        elif tensor == "moving_mean":
            p[name] = (rng.standard_normal(shp, dtype=np.float32) * np.float32(0.1))
        elif tensor == "moving_variance":
            p[name] = rng.uniform(0.5, 1.5, shp).astype(np.float32)
        else:  # pragma: no cover
            raise KeyError(name)
    return p


def save_weights(path: str, params: dict) -> None:
    np.savez(path, **params)


def load_weights_file(path: str) -> dict:
    with np.load(path, allow_pickle=False) as z:
        return {k: np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files}


def check_params(params: dict, n_classes: int, channels: int = 3, arch: str = "fcn_8") -> None:
    want = {"fcn_32": lambda: fcn32_param_shapes(n_classes, channels),
            "fcn_8_vgg": lambda: vgg_param_shapes(n_classes, channels, False),
            "fcn_32_vgg": lambda: vgg_param_shapes(n_classes, channels, True),
            "fcn_8_mobilenet": lambda: mobilenet_param_shapes(n_classes, channels, False),
            "fcn_32_mobilenet": lambda: mobilenet_param_shapes(n_classes, channels, True),
            "fcn_8_resnet50": lambda: resnet50_param_shapes(n_classes, channels, False),
            "fcn_32_resnet50": lambda: resnet50_param_shapes(n_classes, channels, True)}.get(
                arch, lambda: fcn8_param_shapes(n_classes, channels))()
    missing = sorted(set(want) - set(params))
    if missing:
        raise KeyError("weight container lacks tensors: %s" % ", ".join(missing))
    for k, shp in want.items():
        if tuple(params[k].shape) != tuple(shp):
            raise ValueError("tensor %s has shape %s, expected %s" % (k, tuple(params[k].shape), shp))
