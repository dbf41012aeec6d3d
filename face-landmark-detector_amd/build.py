"""Build libflm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m flm_amd.build            # or: python face-landmark-detector_amd/build.py
"""
from __future__ import annotations

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJ_DIR = os.path.join(HERE, "csrc", "build")
LIB_PATH = os.path.join(HERE, "libflm_hip.so")

SOURCES = [
    "flm_api.hip",
    "flm_pack.hip",
    "flm_enc1.hip",
    "flm_igemm.hip",
    "flm_igemm_bf16.hip",
    "flm_conv3_halo.hip",
    "flm_score1x1.hip",
    "flm_tail_bf16.hip",
    "flm_convt.hip",
    "flm_up3_wreg.hip",
    "flm_decode.hip",
    "flm_misc.hip",
    "flm_mobile.hip",
]
# -ffp-contract=off: only the fma() calls written in the sources fuse, so the arithmetic of the
# warp / decode kernels is exactly what their comments (and oracle/) state.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-I", INCLUDE, "-I", CSRC]


# per-source flags.  flm_enc1: the matrix results go to ordinary VGPRs -- the epilogue is the kernel's critical path and
# would otherwise start with one v_accvgpr_read per accumulator register
FILE_FLAGS = {"flm_enc1.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _stamp(paths) -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True, extra_flags=()) -> str:
    """Compile every HIP source and link libflm_hip.so; returns its path."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
              [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    stamp = _stamp(srcs + headers) + " ".join(extra_flags)
    stamp_file = os.path.join(OBJ_DIR, "stamp.txt")
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB_PATH
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), *extra_flags, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp_file, "w") as f:
        f.write(stamp)
    if verbose:
        print("built", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
