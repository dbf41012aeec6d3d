"""ctypes binding of libflm_hip.so (C ABI: include/flm.h).

There is no CPU fallback: if the library is missing or a call fails, the op raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libflm_hip.so")

# enums of include/flm.h
FLM_F32, FLM_BF16 = 0, 1
IN_U8_BGR, IN_F32_RGB = 0, 1
OUT_PROBS, OUT_CLASSMAP, OUT_LANDMARKS, OUT_LOGITS = 0, 1, 2, 3
DECODE_ALL, DECODE_TOPN = 0, 1
NORM_SUB_MEAN, NORM_SUB_AND_DIVIDE, NORM_DIVIDE = 0, 1, 2
ABI_VERSION = 2

EXPORTS = [
    "flm_abi_version", "flm_last_error",
    "flm_fcn8_packed_bytes", "flm_fcn8_pack",
    "flm_fcn32_packed_bytes", "flm_fcn32_pack", "flm_fcn32_workspace_bytes", "flm_fcn32_forward",
    "flm_fcn_packed_bytes", "flm_fcn_pack", "flm_fcn_workspace_bytes", "flm_fcn_forward",
    "flm_forward_opts_init", "flm_fcn_workspace_bytes_opts", "flm_fcn_forward_opts", "flm_fcn8_workspace_offset_opts",
    "flm_fcn8_workspace_bytes", "flm_fcn8_forward", "flm_fcn8_workspace_offset", "flm_fcn8_run_layer",
    "flm_set_tuning", "flm_debug_query", "flm_profile_enable", "flm_profile_filter", "flm_profile_reset", "flm_profile_read", "flm_profile_disable",
    "flm_preprocess",
    "flm_decode_workspace_bytes", "flm_decode",
    "flm_similarity_from_landmarks", "flm_similarity_from_landmarks_scaled", "flm_warp_affine", "flm_crop_resize", "flm_crop_resize_frames",
]


class FlmError(RuntimeError):
    """A call into libflm_hip.so returned a negative status."""


class ConvParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("kernel", "bias", "gamma", "beta", "mean", "var")]


class Fcn8Params(C.Structure):
    _fields_ = [("enc", ConvParams * 5), ("fc6", ConvParams), ("fc7", ConvParams),
                ("score5", ConvParams), ("score4", ConvParams), ("score3", ConvParams),
                ("up5", C.c_void_p), ("up4", C.c_void_p), ("up3", C.c_void_p)]


class ForwardOpts(C.Structure):
    """flm_forward_opts: per-call options that change the workspace layout (include/flm.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("landmark_candidates", C.c_int32), ("candidate_sub_phases", C.c_int32),
                ("candidate_cap_div", C.c_int32)]

    @classmethod
    def make(cls, landmark_candidates=1, candidate_sub_phases=0, candidate_cap_div=1):
        o = cls()
        load().flm_forward_opts_init(C.byref(o))
        o.landmark_candidates = int(landmark_candidates)
        o.candidate_sub_phases = int(candidate_sub_phases)
        o.candidate_cap_div = int(candidate_cap_div)
        return o

    def key(self):
        return (self.landmark_candidates, self.candidate_sub_phases, self.candidate_cap_div)


class FcnParams(C.Structure):
    _fields_ = [("enc", C.POINTER(ConvParams)), ("n_enc", C.c_int), ("fc6", ConvParams), ("fc7", ConvParams),
                ("score5", ConvParams), ("score4", ConvParams), ("score3", ConvParams),
                ("up5", C.c_void_p), ("up4", C.c_void_p), ("up3", C.c_void_p)]


ARCH_FCN8, ARCH_FCN32, ARCH_FCN8_VGG, ARCH_FCN32_VGG, ARCH_FCN8_MOBILENET, ARCH_FCN32_MOBILENET = 0, 1, 2, 3, 4, 5
ARCH_FCN8_RESNET50, ARCH_FCN32_RESNET50 = 6, 7

_lib = None


def _declare(lib):
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    lib.flm_abi_version.restype = i
    lib.flm_abi_version.argtypes = []
    lib.flm_last_error.restype = C.c_char_p
    lib.flm_last_error.argtypes = []
    lib.flm_fcn8_packed_bytes.restype = sz
    lib.flm_fcn8_packed_bytes.argtypes = [i, i]
    lib.flm_fcn8_pack.restype = i
    lib.flm_fcn8_pack.argtypes = [vp, C.POINTER(Fcn8Params), i, i, vp, sz]
    lib.flm_fcn8_workspace_bytes.restype = sz
    lib.flm_fcn8_workspace_bytes.argtypes = [i] * 8
    lib.flm_fcn8_forward.restype = i
    lib.flm_fcn8_forward.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, i, i, f, vp, vp, sz]
    lib.flm_fcn32_packed_bytes.restype = sz
    lib.flm_fcn32_packed_bytes.argtypes = [i, i]
    lib.flm_fcn32_pack.restype = i
    lib.flm_fcn32_pack.argtypes = [vp, C.POINTER(Fcn8Params), i, i, vp, sz]
    lib.flm_fcn32_workspace_bytes.restype = sz
    lib.flm_fcn32_workspace_bytes.argtypes = [i] * 8
    lib.flm_fcn32_forward.restype = i
    lib.flm_fcn32_forward.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, i, i, f, vp, vp, sz]
    lib.flm_fcn_packed_bytes.restype = sz
    lib.flm_fcn_packed_bytes.argtypes = [i, i, i]
    lib.flm_fcn_pack.restype = i
    lib.flm_fcn_pack.argtypes = [vp, i, C.POINTER(FcnParams), i, i, vp, sz]
    lib.flm_fcn_workspace_bytes.restype = sz
    lib.flm_fcn_workspace_bytes.argtypes = [i] * 9
    lib.flm_fcn_forward.restype = i
    lib.flm_fcn_forward.argtypes = [vp, i, vp, vp, i, i, i, i, i, i, i, i, i, f, vp, vp, sz]
    lib.flm_forward_opts_init.restype = None
    lib.flm_forward_opts_init.argtypes = [C.POINTER(ForwardOpts)]
    lib.flm_fcn_workspace_bytes_opts.restype = sz
    lib.flm_fcn_workspace_bytes_opts.argtypes = [i] * 9 + [C.POINTER(ForwardOpts)]
    lib.flm_fcn_forward_opts.restype = i
    lib.flm_fcn_forward_opts.argtypes = [vp, i, vp, vp, i, i, i, i, i, i, i, i, i, f, vp, vp, sz, C.POINTER(ForwardOpts)]
    lib.flm_fcn8_workspace_offset_opts.restype = C.c_int64
    lib.flm_fcn8_workspace_offset_opts.argtypes = [C.c_char_p] + [i] * 8 + [C.POINTER(ForwardOpts)]
    lib.flm_fcn8_workspace_offset.restype = C.c_int64
    lib.flm_fcn8_workspace_offset.argtypes = [C.c_char_p] + [i] * 8
    lib.flm_fcn8_run_layer.restype = i
    lib.flm_fcn8_run_layer.argtypes = [vp, vp, C.c_char_p, vp, vp, i, i, i, i, i]
    lib.flm_profile_filter.restype = i
    lib.flm_profile_filter.argtypes = [C.c_char_p]
    lib.flm_set_tuning.restype = i
    lib.flm_set_tuning.argtypes = [C.c_char_p, i]
    lib.flm_debug_query.restype = i
    lib.flm_debug_query.argtypes = [C.c_char_p, i]
    lib.flm_profile_enable.restype = i
    lib.flm_profile_enable.argtypes = [i]
    lib.flm_profile_reset.restype = i
    lib.flm_profile_reset.argtypes = []
    lib.flm_profile_read.restype = i
    lib.flm_profile_read.argtypes = [i, C.c_char_p, i, C.POINTER(C.c_float)]
    lib.flm_profile_disable.restype = i
    lib.flm_profile_disable.argtypes = []
    lib.flm_preprocess.restype = i
    lib.flm_preprocess.argtypes = [vp, vp, i, i, i, i, vp]
    lib.flm_decode_workspace_bytes.restype = sz
    lib.flm_decode_workspace_bytes.argtypes = [i] * 6
    lib.flm_decode.restype = i
    lib.flm_decode.argtypes = [vp, vp, i, i, i, i, i, i, f, vp, vp, sz]
    lib.flm_similarity_from_landmarks.restype = i
    lib.flm_similarity_from_landmarks.argtypes = [vp, vp, vp, i, i, vp]
    lib.flm_similarity_from_landmarks_scaled.restype = i
    lib.flm_similarity_from_landmarks_scaled.argtypes = [vp, vp, vp, i, i, C.c_double, C.c_double, vp]
    lib.flm_warp_affine.restype = i
    lib.flm_warp_affine.argtypes = [vp, vp, i, i, i, i, vp, vp, i, i]
    lib.flm_crop_resize.restype = i
    lib.flm_crop_resize.argtypes = [vp, vp, i, i, vp, i, vp, i, i]
    lib.flm_crop_resize_frames.restype = i
    lib.flm_crop_resize_frames.argtypes = [vp, vp, C.c_size_t, i, i, i, vp, vp, i, vp, i, i]


def load():
    """Load the HIP library or raise: the product path has no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise FlmError(
            "libflm_hip.so is not built (%s). Build it with `python face-landmark-detector_amd/build.py`; "
            "this package has no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.  Import torch FIRST so the
    # dynamic linker binds this library to that same runtime (same SONAME) -- loading ours first would
    # pull /opt/rocm's copy in beside torch's: two HIP runtimes in one process, and launches on torch's
    # streams / pointers fail with "no ROCm-capable device".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise FlmError("libflm_hip.so lacks symbol %s (stale build?)" % name)
    _declare(lib)
    if lib.flm_abi_version() != ABI_VERSION:
        raise FlmError("libflm_hip.so ABI %d != expected %d" % (lib.flm_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().flm_last_error()
        raise FlmError("%s failed (%d): %s" % (what or "flm call", rc, msg.decode() if msg else "?"))


def require_gpu():
    """The torch device used for HBM allocations and streams; raises without a GPU."""
    import torch
    if not torch.cuda.is_available():
        raise FlmError("no AMD GPU visible to PyTorch-ROCm: the landmark path runs on MI355X only "
                       "(there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr())
