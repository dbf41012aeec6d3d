"""CPU-side checks: the C-ABI library loads and exports every symbol include/flm.h declares
(no compute calls without a GPU), and the host logic that mirrors the reference interface."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import flm_amd  # noqa: F401
from flm_amd import _lib, alignment, distributed, prediction
from flm_amd.networks import LANDMARKS_MODELS, Fcn8Model
from flm_amd import weights as W
from oracle import warp_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "flm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(flm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(lib, s), "libflm_hip.so lacks %s" % s
    assert set(_lib.EXPORTS) == set(syms), set(_lib.EXPORTS) ^ set(syms)
    assert _lib.load().flm_abi_version() == _lib.ABI_VERSION


def test_size_queries_answer_without_a_gpu():
    lib = _lib.load()
    nbytes = lib.flm_fcn8_packed_bytes(68, _lib.FLM_F32)
    # 71,361,868 conv parameters: the blob is that plus padding rows
    assert 4 * 71_361_868 < nbytes < 4 * 71_361_868 * 1.02
    nb16 = lib.flm_fcn8_packed_bytes(68, _lib.FLM_BF16)              # bf16 weights: about half
    assert 2 * 71_361_868 < nb16 < 2 * 71_361_868 * 1.03
    assert lib.flm_fcn8_packed_bytes(68, 7) == 0                       # unknown dtype: says so
    assert lib.flm_fcn8_packed_bytes(1000, _lib.FLM_F32) == 0
    ws = lib.flm_fcn8_workspace_bytes(64, 256, 256, 68, _lib.FLM_F32, _lib.OUT_LANDMARKS, _lib.DECODE_TOPN, 4)
    assert ws > 64 * 264 * 264 * 68 * 4
    assert lib.flm_fcn8_workspace_bytes(1, 250, 256, 68, _lib.FLM_F32, 0, 0, 0) == 0   # H not multiple of 32
    assert b"multiples of 32" in lib.flm_last_error()
    assert lib.flm_decode_workspace_bytes(2, 264, 264, 68, _lib.DECODE_TOPN, 4) > 0


def test_tuning_knobs_and_workspace_layout():
    """flm_set_tuning: documented A/B keys are accepted, unknown keys and bad values fail with a message.  What changes the
    workspace layout is NOT process state: the candidate landmark path is chosen per call by flm_forward_opts, and adds its
    lists to the workspace only where it applies (top-n <= 32, 68 classes, fcn_8)."""
    import ctypes as C
    from flm_amd import _lib
    lib = _lib.load()
    for key, val in ((b"none", 0), (b"bf16_big_tiles", 1), (b"bf16_group_n", 0), (b"bf16_lds_dma", 1), (b"bf16_mfma16", 1), (b"bf16_halo_mfma16", 1), (b"bf16_score1x1", 1), (b"bf16_conv3_halo", 1)):
        assert lib.flm_set_tuning(key, val) == 0, key
    assert lib.flm_set_tuning(b"no_such_knob", 1) != 0 and b"no_such_knob" in lib.flm_last_error()
    assert lib.flm_set_tuning(b"bf16_group_n", 3) != 0
    for key in (b"landmark_candidates", b"candidate_sub_phases", b"candidate_cap_div"):   # moved into the call
        assert lib.flm_set_tuning(key, 1) != 0 and b"flm_forward_opts" in lib.flm_last_error()
    dflt = _lib.ForwardOpts.make()
    assert dflt.struct_size == C.sizeof(_lib.ForwardOpts) == 16 and dflt.key() == (1, 0, 1)
    off_o = _lib.ForwardOpts.make(landmark_candidates=0)
    args = (_lib.ARCH_FCN8, 8, 256, 256, 68, _lib.FLM_F32, _lib.OUT_LANDMARKS, _lib.DECODE_TOPN)
    q = lambda npts, o: lib.flm_fcn_workspace_bytes_opts(*args, npts, C.byref(o) if o is not None else None)
    on, off = q(4, dflt), q(4, off_o)
    assert on > off > 0
    assert q(4, None) == on == lib.flm_fcn_workspace_bytes(*args, 4) == lib.flm_fcn8_workspace_bytes(*args[1:], 4)
    assert q(64, dflt) == q(64, off_o)                      # n > 32: no lists either way
    assert q(4, _lib.ForwardOpts.make(candidate_cap_div=4)) < on
    assert q(25, _lib.ForwardOpts.make(candidate_sub_phases=8)) != q(25, _lib.ForwardOpts.make(candidate_sub_phases=2))
    # all-pixel mode keeps the materialised path
    assert lib.flm_fcn_workspace_bytes_opts(_lib.ARCH_FCN8, 8, 256, 256, 68, _lib.FLM_F32, _lib.OUT_LANDMARKS,
                                            _lib.DECODE_ALL, 0, C.byref(dflt)) < on
    # invalid option structs size nothing and say why
    bad = _lib.ForwardOpts.make()
    bad.struct_size = 4
    assert q(4, bad) == 0 and b"struct_size" in lib.flm_last_error()
    for field, val in (("candidate_sub_phases", 17), ("candidate_sub_phases", -1), ("candidate_cap_div", 0)):
        bad = _lib.ForwardOpts.make()
        setattr(bad, field, val)
        assert q(4, bad) == 0
    assert lib.flm_fcn8_workspace_offset_opts(b"cand_keys", *args[1:], 4, C.byref(off_o)) == -1
    assert lib.flm_fcn8_workspace_offset_opts(b"cand_keys", *args[1:], 4, C.byref(dflt)) > 0


def test_null_arguments_are_rejected_not_dereferenced():
    lib = _lib.load()
    assert lib.flm_decode(None, None, 1, 8, 8, 1, 0, 0, 0.0, None, None, 0) == -1
    assert lib.flm_fcn8_forward(None, None, None, 0, 1, 32, 32, 68, 0, 0, 0, 0, 0.0, None, None, 0) == -1
    assert lib.flm_warp_affine(None, None, 1, 1, 8, 8, None, None, 8, 8) == -1


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    m = LANDMARKS_MODELS["fcn_8"](68, input_height=32, input_width=32)
    with pytest.raises(_lib.FlmError):
        m.predict(np.zeros((1, 32, 32, 3), np.float32))
    with pytest.raises(_lib.FlmError):
        m.load_weights({k: np.zeros(s, np.float32) for k, s in W.fcn8_param_shapes(68).items()})
    from flm_amd.utils import metrics
    with pytest.raises(_lib.FlmError):
        metrics.transfer_target(np.zeros((1, 8, 8, 2), np.float32))


def test_model_object_contract():
    # attributes of networks/utils.py:32-37; defaults of networks/fcn.py:89-90
    m = LANDMARKS_MODELS["fcn_8"](68)
    assert (m.input_height, m.input_width) == (416, 608)
    assert (m.output_height, m.output_width, m.n_classes, m.model_name) == (424, 616, 68, "fcn_8")
    m = LANDMARKS_MODELS["default"](68, input_height=256, input_width=256)
    assert isinstance(m, Fcn8Model) and (m.output_height, m.output_width) == (264, 264)
    with pytest.raises(ValueError):
        LANDMARKS_MODELS["fcn_8"](68, input_height=250, input_width=256)
    rn = LANDMARKS_MODELS["fcn_8_resnet50"](68)                        # defaults 416x608 (fcn.py:167)
    assert (rn.model_name, rn.input_width, rn.output_width, len(rn._enc_layers)) == ("fcn_8_resnet50", 608, 616, 53)
    # every key of the reference's registry (basic_models.py:59-64) builds
    for key in ("fcn_8_resnet50", "fcn_8_mobilenet", "fcn_8_vgg", "default"):
        assert key in LANDMARKS_MODELS
    v = LANDMARKS_MODELS["fcn_8_vgg"](68, input_height=224, input_width=224)   # built, without the download
    assert (v.model_name, v.output_height, len(v._enc_layers)) == ("fcn_8_vgg", 232, 13)
    mb = LANDMARKS_MODELS["fcn_8_mobilenet"](68)                       # defaults 224x224 (fcn.py:181)
    assert (mb.model_name, mb.input_height, mb.output_height, len(mb._enc_layers)) == ("fcn_8_mobilenet", 224, 232, 27)
    v32 = LANDMARKS_MODELS["fcn_32_vgg"](68, input_height=224, input_width=224)
    assert (v32.model_name, v32.output_height) == ("fcn_32_vgg", 256)


def test_weight_container_roundtrip(tmp_path):
    shapes = W.fcn8_param_shapes(5)
    assert shapes["fc6/kernel"] == (7, 7, 256, 4096) and shapes["up3/kernel"] == (16, 16, 5, 5)
    assert sum(int(np.prod(s)) for k, s in W.fcn8_param_shapes(68).items()
               if k.endswith("kernel") or k.endswith("bias")) == 71_361_868
    small = {k: np.full(s, 0.5, np.float32) for k, s in shapes.items() if "fc" not in k}
    with pytest.raises(KeyError):
        W.check_params(small, 5)
    p = W.synth_fcn8_weights(5, seed=3)
    W.check_params(p, 5)
    q = W.synth_fcn8_weights(5, seed=3)
    assert all(np.array_equal(p[k], q[k]) for k in p)
    bad = dict(p)
    bad["up4/kernel"] = np.zeros((4, 4, 5, 6), np.float32)
    with pytest.raises(ValueError):
        W.check_params(bad, 5)


def test_keypts_predict_error_behaviour():
    # prediction.py:166-174
    with pytest.raises(ValueError, match="Both model and checkpoint_path cannot be empty"):
        prediction.keypts_predict()
    m = LANDMARKS_MODELS["fcn_8"](68, input_height=32, input_width=32)
    with pytest.raises(AssertionError):
        prediction.keypts_predict(model=m, inp=None)
    with pytest.raises(AssertionError):
        prediction.keypts_predict(model=m, inp=123)
    with pytest.raises(AssertionError, match="Checkpoint not found"):
        prediction.model_from_checkpoint_path("/nonexistent/ckpt")


def test_find_latest_checkpoint(tmp_path):
    base = str(tmp_path / "ck")
    assert prediction.find_latest_checkpoint(base) is None
    with pytest.raises(ValueError):
        prediction.find_latest_checkpoint(base, fail_safe=False)
    for e in (1, 12, 3):
        open("%s.%05d.npz" % (base, e), "w").close()
    open(base + "_config.json", "w").write(json.dumps({"model_class": "fcn_8", "n_classes": 68}))
    assert prediction.find_latest_checkpoint(base).endswith("ck.00012.npz")


def test_box_maths_matches_oracle():
    rng = np.random.default_rng(0)
    for _ in range(200):
        x0, y0 = int(rng.integers(-20, 1800)), int(rng.integers(-20, 1000))
        face = [x0, y0, x0 + int(rng.integers(1, 400)), y0 + int(rng.integers(1, 400))]
        assert prediction.face_boxes([face])[0] == warp_ref.square_box_ref(face)


def test_shard_range_partitions():
    for total in (0, 1, 7, 64, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [distributed.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        distributed.shard_range(4, 4, 4)


def test_canonical_template():
    t = alignment.canonical_template(68, 256, 256)
    assert t.shape == (68, 2) and t.min() >= 0 and t.max() <= 255
    assert len({tuple(np.round(p, 6)) for p in t}) == 68
    assert np.array_equal(t, alignment.canonical_template(68, 256, 256))
    assert alignment.canonical_template(5, 112, 112).shape == (5, 2)


def test_bench_cpu_baseline_counts_the_cores_it_may_use(monkeypatch, tmp_path):
    """bench.usable_cores: affinity mask cut by the cgroup CPU quota (the GPU box grants 16 CPUs of a 256-thread host;
    sizing the CPU baseline from os.cpu_count() oversubscribed it 16x)."""
    import builtins
    import importlib.util
    spec = importlib.util.spec_from_file_location("flm_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    real_open = builtins.open

    def fake(quota_text):
        f = tmp_path / "cpu.max"
        f.write_text(quota_text)

        def _open(path, *a, **k):
            if path == "/sys/fs/cgroup/cpu.max":
                return real_open(str(f), *a, **k)
            return real_open(path, *a, **k)
        return _open

    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    assert bench.usable_cores() == 16
    monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
    assert bench.usable_cores() == 64          # no quota: the affinity mask, capped at 64 worker threads
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(8)), raising=False)
    assert bench.usable_cores() == 8
