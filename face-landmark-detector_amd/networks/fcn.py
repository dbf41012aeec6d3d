"""FCN-8 landmark model object backed by the HIP library.

Host-side mirror of the Keras model that `fcn_8(n_classes, encoder=vanilla_encoder, ...)`
+ `get_segmentation_model` build (reference networks/fcn.py:10-51,89-126 and
networks/utils.py:6-39): same constructor arguments, same attributes
(`output_width, output_height, n_classes, input_height, input_width, model_name`,
utils.py:32-37), `predict` with the Keras contract ([N,H,W,3] float32 in, [N,H'*W',C]
float32 out, prediction.py:208) and `load_weights` (prediction.py:128).  All arithmetic is
in libflm_hip.so; tensors live in HBM as torch tensors (allocation + streams only).
"""
from __future__ import annotations

import collections
import ctypes as C

import numpy as np

from .. import _lib
from .. import weights as W

_OUT = {"probs": _lib.OUT_PROBS, "classmap": _lib.OUT_CLASSMAP, "landmarks": _lib.OUT_LANDMARKS,
        "logits": _lib.OUT_LOGITS}


def decode_mode_of(n_points: int):
    """n_points < 1 -> all-pixel centroid (utils/metrics.py:58), else top-n (:66)."""
    return (_lib.DECODE_ALL, 0) if n_points < 1 else (_lib.DECODE_TOPN, int(n_points))


class Fcn8Model:
    model_name = "fcn_8"
    _arch = _lib.ARCH_FCN8  # graph selector of the C entry points (enum flm_arch)
    _grid_growth = 8        # output grid = input + 8: (H/8 - 1)*8 + 16, fcn.py:121-124 has no final crop
    _fcn32 = False
    _enc_layers = tuple(("enc%d" % i, True) for i in range(1, 6))   # (tensor prefix, has BatchNorm)

    def __init__(self, n_classes, input_height=416, input_width=608, channels=3, dtype="f32"):
        # defaults as networks/fcn.py:89-90; `dtype` selects the arithmetic of the conv stack:
        # "f32" exact fp32 (parity path) or "bf16" (bf16 operands, fp32 accumulate; BASELINE configs[2])
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        self.dtype = dtype
        self._dt = _lib.FLM_F32 if dtype == "f32" else _lib.FLM_BF16
        if channels != 3:
            raise ValueError("only 3-channel input is built (reference default, fcn.py:90)")
        if input_height % 32 or input_width % 32:
            raise ValueError("input_height/input_width must be multiples of 32 (five 2x2 pools + "
                             "the 8x8 / 16x16 crops of fcn.py:110,118), got %dx%d" % (input_height, input_width))
        self.n_classes = int(n_classes)
        self.input_height = int(input_height)
        self.input_width = int(input_width)
        self.output_height = self.input_height + self._grid_growth
        self.output_width = self.input_width + self._grid_growth
        self._packed = None
        # cached workspaces, least recently used first; one entry is dropped at a time and never one a caller still
        # holds (graphs.CapturedPipeline owns its workspace outright, see new_workspace)
        self._ws = collections.OrderedDict()
        self._ws_cap = 4
        # per-model defaults of the per-call options that change the workspace layout (include/flm.h: flm_forward_opts)
        self.forward_opts = {}
        # the kernels address activations with 32-bit byte offsets: the largest tensor (f1, 64 channels at
        # half resolution) bounds the faces per launch; larger batches are processed in slices
        es = 4 if dtype == "f32" else 2
        big = self.input_height * self.input_width * 64 if self._enc_layers is _VGG_LAYERS else \
            (self.input_height // 2) * (self.input_width // 2) * 64
        self.max_batch = max(1, (2 ** 32 - 1) // (big * es) - 1)

    # ---- weights ------------------------------------------------------------------------------
    def load_weights(self, path_or_params):
        """`model.load_weights(latest_weights)` (prediction.py:128).  Accepts the build's .npz
        container (see weights.py) or a dict of Keras-layout arrays.  Returns None, which the
        caller at prediction.py:130 accepts."""
        params = W.load_weights_file(path_or_params) if isinstance(path_or_params, str) else path_or_params
        self.set_weights(params)
        return None

    def set_weights(self, params: dict):
        import torch
        lib = _lib.load()
        dev = _lib.require_gpu()
        W.check_params(params, self.n_classes, arch=self.model_name)
        held = []

        def up(name):
            t = torch.from_numpy(np.ascontiguousarray(params[name], dtype=np.float32)).to(dev)
            held.append(t)
            return t.data_ptr()

        def conv(name, bn):
            """bn: False, True (BN tensors under the conv's own prefix) or a separate BN layer prefix."""
            cp = _lib.ConvParams()
            kname = name + ("/depthwise_kernel" if name + "/depthwise_kernel" in params else "/kernel")
            cp.kernel = up(kname)
            if name + "/bias" in params:
                cp.bias = up(name + "/bias")
            if bn:
                b = name if bn is True else bn
                cp.gamma = up(b + "/gamma")
                cp.beta = up(b + "/beta")
                cp.mean = up(b + "/moving_mean")
                cp.var = up(b + "/moving_variance")
            return cp

        enc = (_lib.ConvParams * len(self._enc_layers))()
        for i, (name, bn) in enumerate(self._enc_layers):
            enc[i] = conv(name, bn)
        p = _lib.FcnParams()
        p.enc = enc
        p.n_enc = len(self._enc_layers)
        p.fc6 = conv("fc6", False)
        p.fc7 = conv("fc7", False)
        p.score5 = conv("score5", False)
        if not self._fcn32:
            p.score4 = conv("score4", False)
            p.score3 = conv("score3", False)
            p.up5 = up("up5/kernel")
            p.up4 = up("up4/kernel")
            p.up3 = up("up3/kernel")
        else:  # fcn_32: the single 64x64 stride-32 transposed conv travels in the up3 slot
            p.up3 = up("up32/kernel")
        nbytes = lib.flm_fcn_packed_bytes(self._arch, self.n_classes, self._dt)
        if nbytes == 0:
            raise _lib.FlmError("n_classes=%d is outside what the kernels cover" % self.n_classes)
        packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.flm_fcn_pack(_lib.stream_ptr(), self._arch, C.byref(p), self.n_classes, self._dt,
                                    _lib.ptr(packed), nbytes), "flm_fcn_pack")
        torch.cuda.current_stream().synchronize()  # the Keras-layout copies die with `held`
        self._packed = packed

    # ---- forward ------------------------------------------------------------------------------
    def _opts(self, opts):
        """flm_forward_opts for a call: the model's defaults overridden by `opts` (a dict or None)."""
        merged = dict(self.forward_opts)
        merged.update(opts or {})
        return _lib.ForwardOpts.make(**merged)

    def workspace_bytes(self, n, out="probs", n_points=0, opts=None):
        om = _OUT[out]
        dmode, npts = decode_mode_of(n_points) if om == _lib.OUT_LANDMARKS else (0, 0)
        fo = self._opts(opts)
        nbytes = _lib.load().flm_fcn_workspace_bytes_opts(self._arch, n, self.input_height, self.input_width,
                                                          self.n_classes, self._dt, om, dmode, npts, C.byref(fo))
        if nbytes == 0:
            raise _lib.FlmError("workspace query failed: %s" % _lib.load().flm_last_error().decode())
        return int(nbytes)

    def new_workspace(self, n, out="probs", n_points=0, opts=None):
        """A workspace the CALLER owns (pass it as forward_device(..., workspace=ws)): what a captured HIP graph or any
        other holder of raw pointers must use, since cached workspaces may be evicted."""
        import torch
        return torch.empty(self.workspace_bytes(n, out, n_points, opts), dtype=torch.uint8, device=_lib.require_gpu())

    def _workspace(self, n, out_mode, dmode, npts, fo):
        import torch
        key = (n, out_mode, dmode, npts) + fo.key()
        ws = self._ws.get(key)
        if ws is not None:
            self._ws.move_to_end(key)
            return ws
        nbytes = _lib.load().flm_fcn_workspace_bytes_opts(self._arch, n, self.input_height, self.input_width,
                                                          self.n_classes, self._dt, out_mode, dmode, npts, C.byref(fo))
        if nbytes == 0:
            raise _lib.FlmError("workspace query failed: %s" % _lib.load().flm_last_error().decode())
        while len(self._ws) >= self._ws_cap:   # single-entry LRU eviction
            self._ws.popitem(last=False)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=_lib.require_gpu())
        self._ws[key] = ws
        return ws

    def forward_device(self, x, out="probs", n_points=0, thresh=0.0, out_tensor=None, workspace=None, opts=None):
        """One launch sequence on the current stream; everything stays in HBM.

        x: torch CUDA tensor [N,H,W,3], uint8 (raw BGR crop: preprocess fused) or float32
           (already preprocessed, what model.predict receives).
        out: "probs" float32 [N,H'*W',C] | "classmap" int32 [N,H',W'] |
             "landmarks" float64 [N,C,2] | "logits" float32 [N,H',W',C].
        workspace: a caller-owned workspace from `new_workspace` (same n / out / n_points / opts); default: a cached one.
        opts: dict of flm_forward_opts fields (landmark_candidates, candidate_sub_phases, candidate_cap_div).
        """
        import torch
        lib = _lib.load()
        if self._packed is None:
            raise _lib.FlmError("model has no weights: call load_weights()/set_weights() first")
        if x.dim() != 4 or x.shape[1] != self.input_height or x.shape[2] != self.input_width or x.shape[3] != 3:
            raise ValueError("expected [N,%d,%d,3], got %s" % (self.input_height, self.input_width, tuple(x.shape)))
        if not x.is_cuda or not x.is_contiguous():
            raise ValueError("forward_device needs a contiguous CUDA tensor")
        if x.dtype == torch.uint8:
            fmt = _lib.IN_U8_BGR
        elif x.dtype == torch.float32:
            fmt = _lib.IN_F32_RGB
        else:
            raise ValueError("input dtype must be uint8 (BGR) or float32 (preprocessed)")
        n = int(x.shape[0])
        om = _OUT[out]
        if n == 0:  # empty batch: nothing to launch
            oh0, ow0, c0 = self.output_height, self.output_width, self.n_classes
            shp0, dt0 = {_lib.OUT_PROBS: ((0, oh0 * ow0, c0), torch.float32),
                         _lib.OUT_LOGITS: ((0, oh0, ow0, c0), torch.float32),
                         _lib.OUT_CLASSMAP: ((0, oh0, ow0), torch.int32),
                         _lib.OUT_LANDMARKS: ((0, c0, 2), torch.float64)}[om]
            return torch.empty(shp0, dtype=dt0, device=x.device)
        dmode, npts = decode_mode_of(n_points) if om == _lib.OUT_LANDMARKS else (0, 0)
        oh, ow, c = self.output_height, self.output_width, self.n_classes
        shape, dt = {
            _lib.OUT_PROBS: ((n, oh * ow, c), torch.float32),
            _lib.OUT_LOGITS: ((n, oh, ow, c), torch.float32),
            _lib.OUT_CLASSMAP: ((n, oh, ow), torch.int32),
            _lib.OUT_LANDMARKS: ((n, c, 2), torch.float64),
        }[om]
        if out_tensor is None:
            out_tensor = torch.empty(shape, dtype=dt, device=x.device)
        elif tuple(out_tensor.shape) != shape or out_tensor.dtype != dt or not out_tensor.is_contiguous():
            raise ValueError("out_tensor must be contiguous %s %s" % (shape, dt))
        if n > self.max_batch:   # slice the batch (contiguous outputs, same stream)
            if workspace is not None:
                raise ValueError("a caller-owned workspace cannot serve a batch beyond max_batch (%d)" % self.max_batch)
            for lo in range(0, n, self.max_batch):
                hi = min(n, lo + self.max_batch)
                self.forward_device(x[lo:hi], out, n_points, thresh, out_tensor[lo:hi], opts=opts)
            return out_tensor
        fo = self._opts(opts)
        ws = workspace if workspace is not None else self._workspace(n, om, dmode, npts, fo)
        _lib.check(lib.flm_fcn_forward_opts(_lib.stream_ptr(), self._arch, _lib.ptr(self._packed), _lib.ptr(x), fmt, n,
                                             self.input_height, self.input_width, c, self._dt, om, dmode, npts,
                                             float(thresh), _lib.ptr(out_tensor), _lib.ptr(ws), ws.numel(),
                                             C.byref(fo)),
                   "flm_fcn_forward_opts")
        return out_tensor

    def intermediate(self, name, n, out="probs", n_points=0, opts=None, workspace=None):
        """View of a named workspace tensor of the last forward with the same (n, out, n_points, opts) (tests)."""
        import torch
        lib = _lib.load()
        om = _OUT[out]
        dmode, npts = decode_mode_of(n_points) if om == _lib.OUT_LANDMARKS else (0, 0)
        fo = self._opts(opts)
        off = lib.flm_fcn8_workspace_offset_opts(name.encode(), n, self.input_height, self.input_width, self.n_classes,
                                                 self._dt, om, dmode, npts, C.byref(fo))
        if off < 0:
            raise KeyError(name)
        h, w = self.input_height, self.input_width
        bf = self.dtype == "bf16"
        cp = (72 if bf else 68) if self.n_classes == 68 else 16 * ((self.n_classes + 15) // 16)
        shapes = {"f1": (h // 2, w // 2, 64), "f2": (h // 4, w // 4, 128), "f3": (h // 8, w // 8, 256),
                  "f4": (h // 16, w // 16, 256), "f5": (h // 32, w // 32, 256),
                  "fc6": (h // 32, w // 32, 4096), "fc7": (h // 32, w // 32, 4096),
                  "score5": (h // 32, w // 32, cp), "fuse4": (h // 16, w // 16, cp),
                  "seg_feats": (h // 8, w // 8, cp),
                  "probs": (self.output_height * self.output_width, self.n_classes)}
        shp = (n,) + shapes[name]
        ws = workspace if workspace is not None else self._workspace(n, om, dmode, npts, fo)
        cnt = int(np.prod(shp))
        if bf and name in ("f1", "f2", "f3", "f4", "f5", "fc6", "fc7"):   # stored in the operand type
            return ws[off:off + 2 * cnt].view(torch.bfloat16).view(shp).float()
        return ws[off:off + 4 * cnt].view(torch.float32).view(shp)

    def predict(self, x, batch_size=32, verbose=0):
        """Keras `Model.predict` contract (prediction.py:208): numpy in, numpy out, host round trip.
        x: [N,H,W,3] float32 preprocessed (or uint8 BGR crops, then the preprocess is fused)."""
        import torch
        dev = _lib.require_gpu()
        x = np.asarray(x)
        if x.ndim != 4:
            raise ValueError("predict expects [N,H,W,3]")
        if x.shape[0] == 0:
            return np.zeros((0, self.output_height * self.output_width, self.n_classes), np.float32)
        outs = []
        for i in range(0, x.shape[0], batch_size):
            xb = np.ascontiguousarray(x[i:i + batch_size])
            if xb.dtype != np.uint8:
                xb = xb.astype(np.float32, copy=False)
            xd = torch.from_numpy(xb).to(dev)
            outs.append(self.forward_device(xd, "probs").cpu().numpy())
        return np.concatenate(outs, axis=0)


class Fcn32Model(Fcn8Model):
    """fcn_32 (networks/fcn.py:129-150): same encoder and head, one Conv2DTranspose(C, 64x64, stride 32);
    output grid = input + 32.  Weight container: the fcn_8 tensors without score4/score3/up5/up4/up3,
    plus `up32/kernel` (64,64,C,C); `score5/*` is the 1x1 classifier the reference names "seg_feats"."""
    model_name = "fcn_32"
    _arch = _lib.ARCH_FCN32
    _grid_growth = 32
    _fcn32 = True

    def intermediate(self, name, n, out="probs", n_points=0, opts=None, workspace=None):
        raise NotImplementedError("workspace views are exposed for fcn_8 only")


_VGG_LAYERS = tuple(("block%d_conv%d" % (b, c), False)
                    for b, k in ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3)) for c in range(1, k + 1))


class Fcn8VggModel(Fcn8Model):
    """fcn_8 on the VGG16 encoder (networks/fcn.py:153-157, networks/vgg16.py:17-81), built WITHOUT the
    ImageNet download the reference performs by default (`pretrained=None`): 13 conv3x3+ReLU layers
    (tensors `block{b}_conv{c}/kernel|bias`, the Keras layer names), f3/f4/f5 with 256/512/512 channels,
    then the same head."""
    model_name = "fcn_8_vgg"
    _arch = _lib.ARCH_FCN8_VGG
    _enc_layers = _VGG_LAYERS

    def intermediate(self, name, n, out="probs", n_points=0, opts=None, workspace=None):
        raise NotImplementedError("workspace views are exposed for the vanilla fcn_8 only")


class Fcn32VggModel(Fcn8VggModel):
    model_name = "fcn_32_vgg"
    _arch = _lib.ARCH_FCN32_VGG
    _grid_growth = 32
    _fcn32 = True


def fcn_8_vgg(n_classes, input_height=416, input_width=608, channels=3, dtype="f32"):
    """networks/fcn.py:153-157."""
    return Fcn8VggModel(n_classes, input_height=input_height, input_width=input_width, channels=channels, dtype=dtype)


def fcn_32_vgg(n_classes, input_height=416, input_width=608, channels=3, dtype="f32"):
    """networks/fcn.py:160-164."""
    return Fcn32VggModel(n_classes, input_height=input_height, input_width=input_width, channels=channels,
                         dtype=dtype)


_MOBILENET_LAYERS = (("conv1", "conv1_bn"),) + tuple(
    x for i in range(1, 14) for x in (("conv_dw_%d" % i, "conv_dw_%d_bn" % i), ("conv_pw_%d" % i, "conv_pw_%d_bn" % i)))


class Fcn8MobilenetModel(Fcn8Model):
    """fcn_8 on the MobileNet-v1 encoder (networks/fcn.py:181-185, networks/mobilenet.py:59-114; alpha 1),
    built without the ImageNet download.  Tensors carry the Keras layer names: `conv1/kernel`,
    `conv_dw_i/depthwise_kernel` (3,3,C,1), `conv_pw_i/kernel`, and `<layer>_bn/gamma|beta|moving_mean|
    moving_variance`; no biases.  f3/f4/f5 have 256/512/1024 channels."""
    model_name = "fcn_8_mobilenet"
    _arch = _lib.ARCH_FCN8_MOBILENET
    _enc_layers = _MOBILENET_LAYERS

    def __init__(self, n_classes, input_height=224, input_width=224, channels=3, dtype="f32"):
        super().__init__(n_classes, input_height, input_width, channels, dtype)

    def intermediate(self, name, n, out="probs", n_points=0, opts=None, workspace=None):
        raise NotImplementedError("workspace views are exposed for the vanilla fcn_8 only")


class Fcn32MobilenetModel(Fcn8MobilenetModel):
    model_name = "fcn_32_mobilenet"
    _arch = _lib.ARCH_FCN32_MOBILENET
    _grid_growth = 32
    _fcn32 = True


def fcn_8_mobilenet(n_classes, input_height=224, input_width=224, channels=3, dtype="f32"):
    """networks/fcn.py:181-185 (defaults 224x224 as there)."""
    return Fcn8MobilenetModel(n_classes, input_height=input_height, input_width=input_width, channels=channels,
                              dtype=dtype)


def fcn_32_mobilenet(n_classes, input_height=224, input_width=224, channels=3, dtype="f32"):
    """networks/fcn.py:188-192."""
    return Fcn32MobilenetModel(n_classes, input_height=input_height, input_width=input_width, channels=channels,
                               dtype=dtype)


_RESNET_LAYERS = tuple((conv, bn) for conv, bn, _k, _ci, _co in W.resnet50_conv_layers())


class Fcn8Resnet50Model(Fcn8Model):
    """fcn_8 on the ResNet50 encoder (networks/fcn.py:167-171, networks/resnet50.py:122-182), built without the
    ImageNet download: 7x7/s2 stem + 3x3/s2 'valid' max-pool (a 256x256 input gives 63x63 at stage 2) and 16
    bottleneck blocks with fused residual adds; f3/f4/f5 have 512/1024/2048 channels.  Tensors carry the Keras
    layer names (`res3a_branch2b/kernel|bias`, `bn3a_branch2b/gamma|...`).  fp32 or bf16."""
    model_name = "fcn_8_resnet50"
    _arch = _lib.ARCH_FCN8_RESNET50
    _enc_layers = _RESNET_LAYERS

    def intermediate(self, name, n, out="probs", n_points=0, opts=None, workspace=None):
        raise NotImplementedError("workspace views are exposed for the vanilla fcn_8 only")


class Fcn32Resnet50Model(Fcn8Resnet50Model):
    model_name = "fcn_32_resnet50"
    _arch = _lib.ARCH_FCN32_RESNET50
    _grid_growth = 32
    _fcn32 = True


def fcn_8_resnet50(n_classes, input_height=416, input_width=608, channels=3, dtype="f32"):
    """networks/fcn.py:167-171."""
    return Fcn8Resnet50Model(n_classes, input_height=input_height, input_width=input_width, channels=channels,
                             dtype=dtype)


def fcn_32_resnet50(n_classes, input_height=416, input_width=608, channels=3, dtype="f32"):
    """networks/fcn.py:174-178."""
    return Fcn32Resnet50Model(n_classes, input_height=input_height, input_width=input_width, channels=channels,
                              dtype=dtype)


def fcn_32(n_classes, encoder=None, input_height=416, input_width=608, channels=3, dtype="f32"):
    """Signature of networks/fcn.py:129-130 (vanilla encoder only, as fcn_8)."""
    if encoder not in (None, "vanilla", "vanilla_encoder"):
        raise NotImplementedError("only the vanilla encoder (networks/fcn.py:10-51) is built")
    return Fcn32Model(n_classes, input_height=input_height, input_width=input_width, channels=channels, dtype=dtype)


def fcn_8(n_classes, encoder=None, input_height=416, input_width=608, channels=3, dtype="f32"):
    """Signature of networks/fcn.py:89-90.  Only the vanilla conv/BN/ReLU encoder is built
    (`encoder=None` or the string "vanilla"); the ImageNet backbones need a download the
    reference performs at construction time (vgg16.py:76-79 etc.) and are out of scope."""
    if encoder not in (None, "vanilla", "vanilla_encoder"):
        raise NotImplementedError("only the vanilla encoder (networks/fcn.py:10-51) is built")
    return Fcn8Model(n_classes, input_height=input_height, input_width=input_width, channels=channels, dtype=dtype)
