#!/usr/bin/env python3
"""Developer tool: where the time of the bf16 up3 launch goes.  Builds variants of csrc/flm_convt.hip with parts of the
kernel compiled out (-DFLM_ABLATE=<mask>, see the source), links each into libflm_hip.so, and times the layers of the
bf16 batch-512 forward in a child process (results of an ablated build are wrong; only timings mean anything).

    python tools/ablate_up3.py build      # in the build container: writes build_abl/libflm_<mask>.so
    python tools/ablate_up3.py run        # on the GPU box: times every prebuilt variant
"""
import ctypes as C
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "face-landmark-detector_amd")
OUT = os.path.join(ROOT, "build_abl")
MASKS = [int(v) for v in os.environ.get("MASKS", "0,1,3,7,8,16,24,31").split(",")]

CHILD = r'''
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, %r)
import flm_amd
from flm_amd import _lib
_lib.LIB_PATH = %r
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights
lib = _lib.load()
lib.flm_set_tuning(b"up3_cand8", int(os.environ.get("CAND8", "1")))
B = int(os.environ.get("B", "512")); dt = os.environ.get("DTYPE", "bf16")
model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dt)
model.load_weights(synth_fcn8_weights(68, 2))
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
for _ in range(3): model.forward_device(x, "landmarks", n_points=4)
torch.cuda.synchronize()
lib.flm_profile_enable(400); lib.flm_profile_filter(None); lib.flm_profile_reset()
for _ in range(5): model.forward_device(x, "landmarks", n_points=4)
torch.cuda.synchronize()
ms = {}; name = C.create_string_buffer(32); v = C.c_float(); i = 0
while lib.flm_profile_read(i, name, 32, C.byref(v)) == 0:
    ms.setdefault(name.value.decode(), []).append(v.value); i += 1
print("mask %%s: " %% os.environ.get("ABL"), " ".join("%%s %%.3f" %% (k, float(np.median(a))) for k, a in ms.items() if k.startswith("up3") or k in ("decode", "tau")), " total %%.3f" %% sum(float(np.median(a)) for a in ms.values()), flush=True)
'''


def build():
    sys.path.insert(0, ROOT)
    import importlib
    bld = importlib.import_module("face-landmark-detector_amd.build")
    bld.build(force=False)
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(bld.OBJ_DIR, s + ".o") for s in bld.SOURCES if s != "flm_convt.hip"]
    for m in MASKS:
        obj = os.path.join(OUT, "convt_%d.o" % m)
        subprocess.check_call([bld._hipcc(), *bld.FLAGS, "-DFLM_ABLATE=%d" % m, "-c", os.path.join(bld.CSRC, "flm_convt.hip"), "-o", obj])
        subprocess.check_call([bld._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, "libflm_%d.so" % m), obj, *objs])
        os.remove(obj)
        print("built variant", m, flush=True)


def run():
    for m in MASKS:
        lib = os.path.join(OUT, "libflm_%d.so" % m)
        if not os.path.exists(lib):
            continue
        env = dict(os.environ, ABL=str(m))
        subprocess.call([sys.executable, "-c", CHILD % (ROOT, lib)], env=env)


if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
