"""Build hygiene of the HIP sources (CPU only: hipcc cross-compiles gfx950 without a GPU).

Every kernel must compile without scratch memory: a private segment means hipcc left a per-thread
array or a register spill in memory, which on these kernels cost 2-3x (staging registers held in
scratch serialise every load behind `s_waitcnt vmcnt(0)`).  VGPR budgets are pinned where the
occupancy of the design depends on them."""
import concurrent.futures
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "face-landmark-detector_amd", "csrc")
SOURCES = ["flm_igemm.hip", "flm_igemm_bf16.hip", "flm_conv3_halo.hip", "flm_score1x1.hip", "flm_tail_bf16.hip", "flm_convt.hip", "flm_up3_wreg.hip", "flm_enc1.hip", "flm_decode.hip", "flm_misc.hip", "flm_pack.hip", "flm_mobile.hip"]


def _file_flags(src):
    """The per-source flags of the real build (face-landmark-detector_amd/build.py: FILE_FLAGS)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_flm_build", os.path.join(ROOT, "face-landmark-detector_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return list(mod.FILE_FLAGS.get(src, []))


def _asm_metadata(src, tmp):
    out = os.path.join(tmp, src + ".s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", *_file_flags(src),
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    kernels = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)",
                         text):
        kernels[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    return kernels


@pytest.fixture(scope="module")
def metadata(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("asm"))
    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
        res = list(ex.map(lambda s: _asm_metadata(s, tmp), SOURCES))
    merged = {}
    for r in res:
        merged.update(r)
    return merged


def test_no_kernel_uses_scratch(metadata):
    assert len(metadata) >= 40
    bad = {k: v for k, v in metadata.items() if v[0] != 0}
    assert not bad, "kernels with a private segment (scratch): %s" % bad


def test_vgpr_budgets(metadata):
    for name, (_, vgpr) in metadata.items():
        if "igemm_kernel" in name or "igemm_bf16_big_kernel" in name:
            assert vgpr <= 256, (name, vgpr)            # 2 workgroups of 4 waves per CU
        if "convt_kernelILi5ELi17ELb0" in name or "convt_kernelILi5ELi9ELb1" in name:
            assert vgpr <= 256, (name, vgpr)
        if "decode_partial_kernelILi1ELi17ELb0" in name:
            assert vgpr <= 168, (name, vgpr)            # >= 3 waves per SIMD for the streaming decode (n <= 64; the
                                                        # two-register lists of 64 < n <= 128 take 190: 2 waves)
