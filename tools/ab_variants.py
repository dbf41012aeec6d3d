#!/usr/bin/env python3
"""Developer tool: build variants of ONE source of libflm_hip.so with -D<MACRO>=<value> and time the layers of a forward
with each (results of an ablated build may be wrong; only timings mean anything unless the macro's comment says otherwise).

    SRC=flm_igemm.hip MACRO=FLM_IGEMM_VAR VALUES=0,1,2,3 python tools/ab_variants.py build   # build container
    SRC=flm_igemm.hip MACRO=FLM_IGEMM_VAR VALUES=0,1,2,3 [B=64 DTYPE=f32] python tools/ab_variants.py run   # GPU box
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build_abl")
SRC = os.environ.get("SRC", "flm_igemm.hip")
MACRO = os.environ.get("MACRO", "FLM_IGEMM_VAR")
VALUES = [int(v) for v in os.environ.get("VALUES", "0").split(",")]

CHILD = r'''
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, %r)
import flm_amd
from flm_amd import _lib
_lib.LIB_PATH = %r
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights
lib = _lib.load()
for kv in filter(None, os.environ.get("KNOBS", "").split(",")):
    k, v = kv.split("="); _lib.check(lib.flm_set_tuning(k.encode(), int(v)), "set_tuning")
B = int(os.environ.get("B", "64")); dt = os.environ.get("DTYPE", "f32")
model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dt)
model.load_weights(synth_fcn8_weights(68, 2))
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
for _ in range(3): model.forward_device(x, "landmarks", n_points=4)
torch.cuda.synchronize()
lib.flm_profile_enable(1000); lib.flm_profile_filter(None); lib.flm_profile_reset()
for _ in range(int(os.environ.get("STEPS", "8"))): model.forward_device(x, "landmarks", n_points=4)
torch.cuda.synchronize()
ms = {}; name = C.create_string_buffer(32); v = C.c_float(); i = 0
while lib.flm_profile_read(i, name, 32, C.byref(v)) == 0:
    ms.setdefault(name.value.decode(), []).append(v.value); i += 1
keep = os.environ.get("LAYERS", "enc2,enc3,enc4,enc5,fc6,fc7,up3").split(",")
print("%%-4s" %% os.environ.get("VAR"), " ".join("%%s %%.3f" %% (k, float(np.median(a))) for k, a in ms.items() if k in keep),
      " total %%.3f" %% sum(float(np.median(a)) for a in ms.values()), flush=True)
'''


def build():
    sys.path.insert(0, ROOT)
    import importlib
    bld = importlib.import_module("face-landmark-detector_amd.build")
    bld.build(force=False)
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(bld.OBJ_DIR, s + ".o") for s in bld.SOURCES if s != SRC]
    for m in VALUES:
        obj = os.path.join(OUT, "var_%d.o" % m)
        subprocess.check_call([bld._hipcc(), *bld.FLAGS, *bld.FILE_FLAGS.get(SRC, []), "-D%s=%d" % (MACRO, m), "-c",
                               os.path.join(bld.CSRC, SRC), "-o", obj])
        subprocess.check_call([bld._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, "libflm_var_%d.so" % m), obj, *objs])
        os.remove(obj)
        print("built variant", m, flush=True)


def run():
    for rep in range(int(os.environ.get("REPS", "2"))):
        for m in VALUES:
            lib = os.path.join(OUT, "libflm_var_%d.so" % m)
            if os.path.exists(lib):
                subprocess.call([sys.executable, "-c", CHILD % (ROOT, lib)], env=dict(os.environ, VAR=str(m)))


if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
