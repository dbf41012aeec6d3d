#!/usr/bin/env python3
"""Benchmark of the landmark hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--config 4]

One step = one pass of the hot path over one batch of synthetic 256x256 uint8 BGR face crops already resident in HBM:
    fused preprocess + FCN-8 forward -> softmax -> top-n landmark decode (n=4, thresh=0: the reference's as-shipped
    decode) -> similarity fit + alignment warp to 256x256 -> (N > 1) RCCL all-gather of the landmark tensors.

--config 2 (default)  BASELINE.json configs[1]: 64 faces per GPU, fp32 -- the headline (`value`, dtype "f32").
                      At N = 1 the line also carries: cpu_baseline, parity, the bf16 configs[2] run with its own roofline
                      block, and HBM-bound kernel entries.  At N > 1 it also carries `config4` (below) as a side object,
                      so one driver run per N measures both.
--config 3            BASELINE.json configs[2]: 512 faces, bf16 operands / fp32 accumulate, one GPU, as the headline.
--config 4            BASELINE.json configs[3]: 512 bf16 faces per GPU (4096 at 8) + all-gather, as the headline.
--config 5            BASELINE.json configs[4]: 1080p multi-face stream -- per GPU 64 synthetic 1920x1080 frames with 1..16
                      face boxes each (sides 96..400 px): box maths + crop + resize to 256x256 + landmarks + alignment,
                      16 frames per launch sequence; one step = one pass over a rank's frames; frames are independent, no
                      collective.  `value` is faces/s, `frames_per_s` rides along.
Ranks are weak-scaled: every rank runs its own batch; the only collective is the landmark gather.  ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import concurrent.futures
import ctypes as C
import json
import os
import statistics
import sys
import time

# the pool's host driver only supports dmabuf IPC: RCCL fails with hipIpcGetMemHandle errors without this (set before HIP starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = FP32 vector peak
PEAK_BF16_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)
DECODE_BYTES_PER_FACE = 18957312 + 544     # fp32 [264,264,68] read + [68,2] f32-sized written
WARP_BYTES_PER_FACE = 196608 + 786432      # u8 source + f32 destination


# ---- work accounting ---------------------------------------------------------------------------------------------
def _same_conv_fraction(size, k):
    """Fraction of the k x k taps of a 'same' convolution on a size x size map that fall on real pixels."""
    pad = k // 2
    valid = sum(sum(1 for t in range(k) if 0 <= y + t - pad < size) for y in range(size))
    return (valid / (size * k)) ** 2


def layer_work(h=256, c=68):
    """{layer: (dense, useful)} GFLOP per face, 2*MAC.  dense = SURVEY 8(d)'s count (zero-padded taps included: 17.844
    in total); useful = the same without the products whose activation operand is zero padding (fc6's 7x7 'same' on
    8x8: 61.7 %; the 3x3 layers: 91.8-99.5 %).  `roofline.frac` prices USEFUL work: a kernel that skips padding taps
    (fc6 does) must not read above 1."""
    g = {}
    chans = [(3, 64), (64, 128), (128, 256), (256, 256), (256, 256)]
    s = h
    for i, (ci, co) in enumerate(chans):
        dense = 2.0 * s * s * 9 * ci * co / 1e9
        g["enc%d" % (i + 1)] = (dense, dense * _same_conv_fraction(s, 3))
        s //= 2
    d6 = 2.0 * s * s * 49 * 256 * 4096 / 1e9
    g["fc6"] = (d6, d6 * _same_conv_fraction(s, 7))
    for name, k, n in (("fc7", 4096, 4096), ("score5", 4096, c)):
        g[name] = (2.0 * s * s * k * n / 1e9,) * 2
    g["score4"] = (2.0 * (2 * s) ** 2 * 256 * c / 1e9,) * 2
    g["score3"] = (2.0 * (4 * s) ** 2 * 256 * c / 1e9,) * 2
    g["up5"] = (2.0 * s * s * 16 * c * c / 1e9,) * 2
    g["up4"] = (2.0 * (2 * s) ** 2 * 16 * c * c / 1e9,) * 2
    g["up3"] = (2.0 * (4 * s) ** 2 * 256 * c * c / 1e9,) * 2
    return g


WORK = layer_work()
GFLOP_PER_FACE = sum(v[0] for v in WORK.values())          # 17.844
USEFUL_GFLOP_PER_FACE = sum(v[1] for v in WORK.values())   # 15.18


def fc6_issued_gflop(batch):
    """MACs the fp32 fc6 actually issues at this batch: position-major 128-row tiles skip the filter taps that see
    only zero padding for the whole tile (csrc/flm_igemm.hip); returned as 2*MAC in GFLOP."""
    n, h, w, kh, pad, cin, cout = batch, 8, 8, 7, 3, 256, 4096
    m_total = n * h * w
    taps = 0
    for t in range((m_total + 127) // 128):
        lo, hi = t * 128, min(t * 128 + 128, m_total) - 1
        valid = set()
        for p in range(lo // n, hi // n + 1):
            y, x = divmod(p, w)
            for ky in range(kh):
                if not 0 <= y + ky - pad < h:
                    continue
                for kx in range(kh):
                    if 0 <= x + kx - pad < w:
                        valid.add((ky, kx))
        taps += len(valid)
    return 2.0 * taps * 128 * cin * cout / 1e9


def load_traffic(name, layer):
    """HBM bytes per launch of a layer's kernel from a COMMITTED PMC summary (None when absent): profiles/<name> =
    {layer: FETCH_SIZE*2 + WRITE_SIZE in bytes}, see profiles/README.md.  Not measured by this run."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f).get(layer)
    except (OSError, ValueError):
        return None


# decode side of the step: the landmark selection ("up3_sub" + "tau" = its threshold pass over a sample of the map) and
# the gated fallback launches; everything else is the FCN forward whose FLOPs SURVEY 8(d) counts
DECODE_KEYS = ("decode", "up3_sub", "tau", "decode_fallback", "up3_fallback")
RECORDS_PER_STEP = 24   # launches one step brackets at most
LAYER_PASS_STEPS = 3    # steps of the separate per-layer timing pass


def read_profile(lib):
    """Per-layer mean duration (ms) of the launches recorded since the last reset."""
    layer_ms = {}
    name = C.create_string_buffer(32)
    ms = C.c_float()
    i = 0
    while lib.flm_profile_read(i, name, 32, C.byref(ms)) == 0:
        layer_ms.setdefault(name.value.decode(), []).append(ms.value)
        i += 1
    return {k: sum(v) / len(v) for k, v in layer_ms.items()}


# ---- the harness (shared with tests/test_bench_harness_gloo.py, which injects a CPU step) --------------------------
def timed_region(step, steps, warmup, world=1, device=None, settle_s=0.0, before_timed=None):
    """W untimed warm-up steps (+ `settle_s` seconds more of them, so the clocks have settled when a short timed region
    starts), then EXACTLY `steps` steps bracketed by barrier + device synchronise on both sides; returns the wall time in
    seconds, MAX over ranks, the last step's result, and every rank's own ms per step (a straggler shows as the max)."""
    import torch
    import torch.distributed as dist
    on_gpu = device is not None and getattr(device, "type", "cpu") == "cuda"

    def fence():
        # a pipelined step (pipeline_gather at N > 1) leaves its last all-gather in flight: it belongs to the region
        # that is closing; returns that gather's result (None otherwise)
        last = step.drain() if hasattr(step, "drain") else None
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()
        return last

    out = None
    for _ in range(warmup):
        out = step()
    if settle_s > 0:
        # a COUNT of extra steps agreed between the ranks (a time-based loop would run a different number of
        # collectives on each rank): one probe step gives the step time, the slowest rank's count is taken
        if on_gpu:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = step()
        if on_gpu:
            torch.cuda.synchronize()
        extra = min(500, int(settle_s / max(time.perf_counter() - t0, 1e-5)))
        if world > 1:
            t = torch.tensor([extra], dtype=torch.int64, device=device if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            extra = int(t.item())
        for _ in range(extra):
            out = step()
    if before_timed is not None:
        before_timed()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    last = fence()
    dt = time.perf_counter() - t0
    if last is not None:
        out = (last,) + tuple(out[1:])   # the last step's own gather
    mine = [1e3 * dt / max(steps, 1)]
    if world > 1:
        every = torch.zeros(world, dtype=torch.float64, device=device if on_gpu else "cpu")
        dist.all_gather_into_tensor(every, torch.tensor([dt], dtype=torch.float64, device=every.device))
        mine = [1e3 * float(v) / max(steps, 1) for v in every.cpu()]
        dt = max(float(v) for v in every.cpu())
    per_rank = {"min": min(mine), "median": statistics.median(mine), "max": max(mine), "ranks": mine}
    return dt, out, per_rank


def time_gather(world, device, rows, classes=68, reps=20):
    """The landmark all-gather ALONE, after the timed region: mean ms per call on the slowest rank (None at N = 1)."""
    import torch
    import torch.distributed as dist
    from flm_amd import distributed
    if world <= 1 or not dist.is_initialized():
        return None
    on_gpu = device is not None and getattr(device, "type", "cpu") == "cuda"
    lm = torch.zeros((rows, classes, 2), dtype=torch.float64, device=device if on_gpu else "cpu")
    for _ in range(3):
        distributed.all_gather_landmarks(lm, rows * world)
    dist.barrier()
    if on_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        distributed.all_gather_landmarks(lm, rows * world)
    if on_gpu:
        torch.cuda.synchronize()
    t = torch.tensor([1e3 * (time.perf_counter() - t0) / reps], dtype=torch.float64, device=lm.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def describe_collective(world, device=None, payload_rows=0, classes=68, dtype_name="f64"):
    """What a reader needs to check that the collective really ran over `world` ranks: the world size torch.distributed
    reports after init, the backend, the RCCL version, every rank's device name."""
    import torch
    import torch.distributed as dist
    if world <= 1 or not dist.is_initialized():
        return {"op": "none", "ranks_seen": 1, "backend": None, "rccl_version": None,
                "devices": [torch.cuda.get_device_name(device)] if device is not None and device.type == "cuda" else ["cpu"]}
    on_gpu = device is not None and device.type == "cuda"
    mine = "%s (cuda:%d)" % (torch.cuda.get_device_name(device), device.index) if on_gpu else "cpu"
    names = [None] * dist.get_world_size()
    dist.all_gather_object(names, mine)
    ver = None
    if on_gpu:
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001 -- version probe only
            ver = "unknown"
    es = 8 if dtype_name == "f64" else 4
    return {"op": "all_gather_into_tensor", "ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
            "rccl_version": ver, "devices": names,
            "payload": "[%d,%d,2] %s per rank = %d B" % (payload_rows, classes, dtype_name, payload_rows * classes * 2 * es),
            "payload_deviation": "north_star / SURVEY 8(e) name an fp32 payload ([512,68,2] = 278,528 B per rank); this "
                                 "build gathers float64 (twice the bytes): it is what the reference's decode returns "
                                 "(utils/metrics.py:102-109) and keeps the gathered coordinates the single-GPU bits; the "
                                 "exchange is latency-bound either way"}


def pipeline_gather(produce, world, total):
    """step() for N ranks: `produce()` queues this rank's batch and returns (landmarks [b, C, 2], aux); its all-gather is
    queued behind it (RCCL's own stream) and the PREVIOUS step's gather is waited for only then -- the exchange of batch
    i hides behind the kernels of batch i + 1 instead of stalling the stream once per batch.  step() returns (the
    previous step's gathered landmarks or None, aux); step.drain() waits for the one still in flight (timed_region calls
    it before closing a region, so K timed steps contain K complete exchanges).  N = 1: no exchange, (landmarks, aux)."""
    from flm_amd import distributed
    pending = [None]

    def step():
        lm, aux = produce()
        if world <= 1:
            return lm, aux
        nxt = distributed.all_gather_landmarks_async(lm, total)
        full = pending[0].wait() if pending[0] is not None else None
        pending[0] = nxt
        return full, aux

    def drain():
        full = pending[0].wait() if pending[0] is not None else None
        pending[0] = None
        return full
    step.drain = drain
    return step


class Workload:
    """One configuration of the step on this rank: model, resident crops, template."""

    def __init__(self, dtype, batch, rank, n_points, align=True, seed=1):
        import numpy as np
        import torch
        from flm_amd import _lib, alignment
        from flm_amd.networks import LANDMARKS_MODELS
        from flm_amd.weights import synth_fcn8_weights
        self.dtype, self.batch, self.n_points, self.align = dtype, batch, n_points, align
        self.H = self.W = 256
        self.dev = _lib.require_gpu()
        self.model = LANDMARKS_MODELS["fcn_8"](68, input_height=self.H, input_width=self.W, dtype=dtype)
        self.model.load_weights(synth_fcn8_weights(68, seed=2))
        rng = np.random.default_rng(seed + rank)
        self.crops = torch.from_numpy(rng.integers(0, 256, (batch, self.H, self.W, 3), dtype=np.uint8)).to(self.dev)
        self.tmpl = torch.from_numpy(alignment.canonical_template(68, self.H, self.W)).to(self.dev)
        self.scale = (self.W / self.model.output_width, self.H / self.model.output_height)

    def make_step(self, world, total):
        from flm_amd import alignment

        def produce():
            lm = self.model.forward_device(self.crops, "landmarks", n_points=self.n_points, thresh=0.0)
            aligned = None
            if self.align:
                aligned, _m = alignment.align_device(self.crops, lm, self.tmpl, self.H, self.W, self.scale)
            return lm, aligned
        return pipeline_gather(produce, world, total)


class StreamWorkload:
    """BASELINE configs[4] on this rank: 1080p frames resident in HBM, 1..16 boxes per frame, `group` frames per launch
    sequence (the detector that would supply the boxes does not exist in the reference, prediction.py:99,103)."""

    def __init__(self, dtype, rank, n_points, n_frames=64, group=64):
        import numpy as np
        import torch
        from flm_amd import _lib, alignment
        from flm_amd.networks import LANDMARKS_MODELS
        from flm_amd.weights import synth_fcn8_weights
        self.dtype, self.n_points, self.group, self.n_frames = dtype, n_points, group, n_frames
        self.dev = _lib.require_gpu()
        self.model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dtype)
        self.model.load_weights(synth_fcn8_weights(68, seed=2))
        rng = np.random.default_rng(5 + rank)
        # a ring of 8 frames in one allocation: the faces of a whole group are cut in one launch
        self.frames = torch.from_numpy(rng.integers(0, 256, (8, 1080, 1920, 3), dtype=np.uint8)).to(self.dev)
        self.faces = []
        for _ in range(n_frames):
            fb = []
            for _ in range(int(rng.integers(1, 17))):
                side = int(rng.integers(96, 401))
                x0, y0 = int(rng.integers(0, 1920 - side)), int(rng.integers(0, 1080 - side))
                fb.append((x0, y0, x0 + side, y0 + side))
            self.faces.append(fb)
        self.batch = sum(len(f) for f in self.faces)   # faces per step
        self.tmpl = torch.from_numpy(alignment.canonical_template(68, 256, 256)).to(self.dev)
        self.scale = (256 / self.model.output_width, 256 / self.model.output_height)

    def make_step(self, world, total):
        import torch
        from flm_amd import alignment, prediction

        def step():
            lm = None
            for f0 in range(0, self.n_frames, self.group):
                fr = range(f0, min(f0 + self.group, self.n_frames))
                crops, _ = prediction.crop_frames_device(self.frames, [self.faces[f] for f in fr], 256, 256,
                                                         frame_index=[f % self.frames.shape[0] for f in fr])
                lm = self.model.forward_device(crops, "landmarks", n_points=self.n_points)
                alignment.align_device(crops, lm, self.tmpl, 256, 256, self.scale)
            return lm, None
        return step


def measure(lib, wl, steps, warmup, world, roof_layer, settle_s):
    """Timed region of one workload with the roofline kernel's launches bracketed by HIP events on the launch stream,
    then the per-layer pass.  Returns (seconds, roofline-layer ms, {layer: ms})."""
    import torch
    from flm_amd import _lib
    step = wl.make_step(world, wl.batch * world)
    _lib.check(lib.flm_profile_enable(max(steps, LAYER_PASS_STEPS) * RECORDS_PER_STEP + 64), "flm_profile_enable")
    _lib.check(lib.flm_profile_filter(roof_layer.encode()), "flm_profile_filter")

    dt, _, per_rank = timed_region(step, steps, warmup, world, wl.dev, settle_s, before_timed=lib.flm_profile_reset)
    roof_ms = read_profile(lib).get(roof_layer, float("nan"))
    _lib.check(lib.flm_profile_filter(None), "flm_profile_filter")
    lib.flm_profile_reset()
    for _ in range(LAYER_PASS_STEPS):
        step()
    if hasattr(step, "drain"):
        step.drain()
    torch.cuda.synchronize()
    layers = read_profile(lib)
    lib.flm_profile_disable()
    return dt, roof_ms, layers, per_rank


def forward_summary(layers, batch, peak):
    fwd_ms = sum(v for k, v in layers.items() if k not in DECODE_KEYS)
    table = {}
    for k, ms in layers.items():
        if k in WORK:
            table[k] = {"ms": ms, "useful_tflops": WORK[k][1] * batch / ms, "frac": WORK[k][1] * batch / ms / peak}
    return {"gflop_per_face_useful": USEFUL_GFLOP_PER_FACE, "gflop_per_face_dense": GFLOP_PER_FACE, "ms": fwd_ms,
            "tflops": USEFUL_GFLOP_PER_FACE * batch / fwd_ms,
            "frac_of_mfma_peak": USEFUL_GFLOP_PER_FACE * batch / fwd_ms / peak,
            "tflops_dense": GFLOP_PER_FACE * batch / fwd_ms,
            "frac_dense": GFLOP_PER_FACE * batch / fwd_ms / peak,
            "faces_per_s_forward_only": 1e3 * batch / fwd_ms, "layer_ms": layers, "layer_roofline": table,
            "note": "tflops / frac price USEFUL 2*MAC (no zero-padding products, %.3f GFLOP per face); *_dense use "
                    "SURVEY 8(d)'s 17.844.  layer_ms: separate pass of %d steps after the timed region, every launch "
                    "bracketed by HIP events" % (USEFUL_GFLOP_PER_FACE, LAYER_PASS_STEPS)}


def mfma_roofline(layer, kernel, ms, batch, peak, traffic_file, extra=None):
    dense, useful = WORK[layer]
    r = {"bound": "mfma", "kernel": kernel, "achieved": useful * batch / ms, "peak": peak, "unit": "TFLOP/s",
         "frac": useful * batch / ms / peak, "traffic": load_traffic(traffic_file, layer),
         "flop_per_launch": useful * 1e9 * batch, "avg_launch_ms": ms,
         "frac_dense": dense * batch / ms / peak,
         "traffic_source": "committed profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE "
                           "doubled per MI355X_MICROARCH.md), not measured by this run" % traffic_file,
         "note": "achieved = useful 2*MAC per launch (products with zero padding excluded) / mean launch duration from "
                 "HIP events on the launch stream inside the timed region; frac_dense uses SURVEY 8(d)'s dense count"}
    if extra:
        r.update(extra)
    return r


def hbm_kernels(lib, dev, only=None):
    """The HBM-bound kernels on their own (outside any timed region): standalone decode of materialised heatmaps and the
    alignment warp, algorithmic bytes / mean launch duration / 8 TB/s.  `only`: restrict to the entries whose key is listed
    (tools/bench_hbm.py under rocprofv3: one batch size per profiled run)."""
    import torch
    from flm_amd import alignment
    from flm_amd.utils.metrics import decode_device
    out = {}

    def timeit(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for nb in (64, 512):
        if only and "decode_top4_b%d" % nb not in only:
            continue
        hm = torch.rand((nb, 264, 264, 68), dtype=torch.float32, device=dev)
        ms = timeit(lambda: decode_device(hm, 4, 0.0))
        gbs = DECODE_BYTES_PER_FACE * nb / ms / 1e6
        out["decode_top4_b%d" % nb] = {"bound": "hbm", "kernel": "decode_partial_dma_kernel + decode_merge_kernel (flm_decode, "
                                       "float32 [%d,264,264,68] -> landmarks)" % nb, "achieved": gbs, "peak": PEAK_HBM_GBS,
                                       "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "avg_launch_ms": ms,
                                       "bytes_per_launch": DECODE_BYTES_PER_FACE * nb,
                                       "traffic": load_traffic("traffic_latest.json", "decode_standalone_b%d" % nb)}
        del hm
    for nb in (64, 512):
        if only and "warp_b%d" % nb not in only:
            continue
        src = torch.randint(0, 256, (nb, 256, 256, 3), dtype=torch.uint8, device=dev)
        m = torch.tensor([[0.98, 0.05, 2.0], [-0.05, 0.98, 3.0]], dtype=torch.float32, device=dev).repeat(nb, 1, 1).contiguous()
        dst = torch.empty((nb, 256, 256, 3), dtype=torch.float32, device=dev)
        ms = timeit(lambda: alignment.warp_device(src, m, 256, 256, out=dst))
        gbs = WARP_BYTES_PER_FACE * nb / ms / 1e6
        out["warp_b%d" % nb] = {"bound": "hbm", "kernel": "warp_u8_rows_kernel (flm_warp_affine, u8 [%d,256,256,3] -> f32)" % nb,
                                "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                "avg_launch_ms": ms, "bytes_per_launch": WARP_BYTES_PER_FACE * nb,
                                "traffic": load_traffic("traffic_latest.json", "warp_b%d" % nb)}
        del src, dst
    return out


# ---- CPU baseline --------------------------------------------------------------------------------------------------
def usable_cores():
    """CPUs this process may really use: the affinity mask, cut by a cgroup CPU quota if one is set (the GPU box hands a
    16-CPU share of a 256-thread host: os.cpu_count() alone oversubscribes the baseline 16x)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:            # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:                                                  # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 64))


def cpu_baseline(n_faces, n_points, seed, reps=5, warmups=2):
    """The oracle (kind "port": the build's CPU restatement of prediction.py's path) timed on the host cores: preprocess +
    FCN-8 forward + softmax + top-n decode of `n_faces` crops per repetition; BASELINE.md section 3's protocol (2 warm-ups,
    median of >= 5 repetitions) on a sample bounded to ~15 s of CPU work (the first `n_faces` crops of the headline batch)."""
    import numpy as np
    import torch
    from flm_amd.weights import synth_fcn8_weights
    from oracle import decode_ref, fcn_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    params = synth_fcn8_weights(68, seed=2)
    rng = np.random.default_rng(seed)
    crops = rng.integers(0, 256, (64, 256, 256, 3), dtype=np.uint8)[:n_faces]

    def decode_face(hm_one):  # the reference's per-landmark argsort loop (utils/metrics.py:66-77), one face
        with np.errstate(all="ignore"):
            return decode_ref.transfer_target_ref(hm_one, 0, n_points)

    def run(c):
        x = np.stack([fcn_ref.get_image_array_ref(i) for i in c])
        pr = fcn_ref.fcn8_predict_ref(x, params).reshape(len(c), 264, 264, 68)
        # faces are independent: one decode per worker thread (numpy's sort releases the GIL)
        with concurrent.futures.ThreadPoolExecutor(max_workers=cores) as ex:
            return np.concatenate(list(ex.map(decode_face, [pr[i:i + 1] for i in range(len(c))])))

    for _ in range(warmups):
        run(crops)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        lm = run(crops)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    # the yardstick of the parity block (outside the timing): the same crops through the FLOAT64 oracle, probabilities
    # rounded to float32 (what model.predict returns) and decoded by the reference's arithmetic
    p64 = fcn_ref.fcn8_predict_ref(np.stack([fcn_ref.get_image_array_ref(i) for i in crops]), params, torch.float64)
    with concurrent.futures.ThreadPoolExecutor(max_workers=cores) as ex:
        lm64 = np.concatenate(list(ex.map(decode_face, [p64[i:i + 1].astype(np.float32).reshape(1, 264, 264, 68)
                                                        for i in range(len(crops))])))
    gap = np.stack([decode_ref.topn_gap_rel(p64[i], n_points) for i in range(len(crops))]) if n_points >= 1 else None
    del p64
    cpu_baseline.float64 = (lm64, gap)
    return {"value": n_faces / med, "unit": "faces/s", "cores": cores, "kind": "port",
            "sample": "%d synthetic 256x256 crops per repetition through oracle/ (torch-CPU fp32 forward on %d threads + "
                      "numpy top-%d decode, one face per thread); %d warm-ups, median of %d repetitions (%.2f s each, "
                      "min %.2f max %.2f)" % (n_faces, cores, n_points, warmups, reps, med, min(times), max(times))}, lm, crops


# ---- main ----------------------------------------------------------------------------------------------------------
def bf16_side_object(lib, args, world, rank, tag):
    """BASELINE configs[2] (N = 1) / configs[3] (N > 1): 512 bf16 faces per rank, same step, own timed region."""
    wl = Workload("bf16", args.bf16_batch, rank, args.n_points, not args.no_align, seed=3)
    steps = max(5, args.steps)
    dt, up3_ms, layers, per_rank = measure(lib, wl, steps, max(2, args.warmup), world, "up3", args.settle_ms / 1e3)
    total = wl.batch * world
    fwd = forward_summary(layers, wl.batch, PEAK_BF16_TFLOPS)
    obj = {"workload": "%s: %d faces per GPU x %d GPU(s), 256x256 crops, bf16 operands / fp32 accumulate, same step as "
                       "the headline%s" % (tag, wl.batch, world, " + all-gather of the landmarks" if world > 1 else ""),
           "dtype": "bf16", "n_gpus": world, "faces_per_s": total * steps / dt, "ms_per_step": 1e3 * dt / steps,
           "steps": steps, "per_rank_ms_per_step": per_rank, "gather_ms": time_gather(world, wl.dev, wl.batch),
           "roofline": mfma_roofline("up3", "up3_cand8_kernel<bf16, 8 waves> (up3: Conv2DTranspose 16x16 s8 68->68 on 32x32, "
                                     "softmax + candidate keys in the epilogue)", up3_ms, wl.batch, PEAK_BF16_TFLOPS,
                                     "traffic_bf16_latest.json"),
           "forward": fwd}
    return obj, wl


def stream_config(lib, args, rank, world, dev):
    """--config 5: the stream as the headline line (no roofline block of its own: its launches have a different batch per
    sequence; the kernels are the ones the other configurations price)."""
    wl = StreamWorkload(args.stream_dtype, rank, args.n_points, group=args.stream_group)
    step = wl.make_step(world, 0)
    dt, _, per_rank = timed_region(step, args.steps, args.warmup, world, dev, args.settle_ms / 1e3)
    import torch
    import torch.distributed as dist
    faces = torch.tensor([wl.batch], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(faces)
    coll = describe_collective(world, dev, 0, 68)
    if rank == 0:
        total = int(faces.item())
        print(json.dumps({
            "metric": "faces/sec (whole node), 1080p multi-face stream: box maths + crop/resize to 256x256 + FCN-8 landmarks "
                      "(top-%d) + similarity/alignment warp" % args.n_points,
            "value": total * args.steps / dt, "unit": "faces/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.stream_dtype, "data": "synthetic", "per_rank_ms_per_step": per_rank,
            "frames_per_s": wl.n_frames * world * args.steps / dt,
            "config": {"workload": "BASELINE configs[4]: per GPU %d synthetic 1920x1080x3 uint8 frames resident in HBM, 1..16 "
                                   "boxes each (sides 96..400 px, %d faces on rank 0), %d frames per launch sequence, "
                                   "fcn_8(68) %s, decode top-%d, align to 256x256; no detector exists in the reference "
                                   "(prediction.py:99,103): boxes are synthetic" % (wl.n_frames, wl.batch, wl.group,
                                                                                    args.stream_dtype, args.n_points),
                       "faces_per_step_all_ranks": total, "frames_per_gpu_per_step": wl.n_frames, "parallelism": "dp%d" % world,
                       "collective": dict(coll, op="none (frames are independent; the device names were gathered)")},
            "roofline": None, "cpu_baseline": None}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5),
                    help="BASELINE.json configs[config-1] as the headline: 2 = 64 fp32 faces per GPU (default), 3 = 512 "
                         "bf16 faces on one GPU, 4 = 512 bf16 faces per GPU + all-gather, 5 = the 1080p multi-face stream")
    ap.add_argument("--stream-dtype", default="f32", choices=("f32", "bf16"), help="arithmetic of --config 5")
    ap.add_argument("--batch", type=int, default=0, help="faces per GPU per step (0 = the configuration's: 64 or 512)")
    ap.add_argument("--n-points", type=int, default=4)
    ap.add_argument("--cpu-faces", type=int, default=16, help="faces per repetition of the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-align", action="store_true")
    ap.add_argument("--bf16-batch", type=int, default=512,
                    help="with --config 2: also time the bf16 configuration (this many faces per GPU per step) and report "
                         "it as a side object (`bf16_config3` at N = 1, `config4` at N > 1); 0 = skip")
    ap.add_argument("--f32-big-batch", type=int, default=512,
                    help="with --config 2 at N = 1: also time this many fp32 faces in one batch (side object "
                         "`f32_batch512`); 0 = skip")
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="extra untimed warm-up before the timed region so a short region starts at settled clocks")
    ap.add_argument("--no-hbm-kernels", action="store_true")
    ap.add_argument("--stream-group", type=int, default=64,
                    help="--config 5: frames whose faces share one launch sequence (crop / resize, landmarks, alignment); "
                         "64 = the whole step (about 500 faces per sequence), 16 = a shorter pipeline (about 130)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend for N > 1: nccl (= RCCL, the measured configuration) or gloo (a dry "
                         "run of the multi-rank flow on a box with fewer GPUs than ranks: ranks share devices, the "
                         "gather goes through the host; its numbers mean nothing)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import flm_amd  # noqa: F401
    from flm_amd import _lib, distributed

    rank, local_rank, world = distributed.env_world()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                         "--nproc-per-node %d ... bench.py --gpus %d" % (args.gpus, args.gpus, args.gpus))
    if args.config == 3 and world > 1:
        raise SystemExit("--config 3 is the one-GPU bf16 run; use --config 4 for N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    if world > 1 and args.backend == "nccl" and ndev < world:
        raise SystemExit("--gpus %d needs %d visible GPUs for RCCL, found %d" % (world, world, ndev))
    torch.cuda.set_device(local_rank % ndev if world > 1 else 0)
    if world > 1:
        distributed.init_process_group(args.backend)
    dev = torch.device("cuda", torch.cuda.current_device())
    lib = _lib.load()

    if args.config == 5:
        stream_config(lib, args, rank, world, dev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    bf16_head = args.config in (3, 4)
    dtype = "bf16" if bf16_head else "f32"
    B = args.batch or (512 if bf16_head else 64)
    CLS = 68
    wl = Workload(dtype, B, rank, args.n_points, not args.no_align, seed=3 if bf16_head else 1)
    roof_layer = "up3" if bf16_head else "fc6"
    peak = PEAK_BF16_TFLOPS if bf16_head else PEAK_F32_TFLOPS
    dt, roof_ms, layer_avg, per_rank = measure(lib, wl, args.steps, args.warmup, world, roof_layer, args.settle_ms / 1e3)
    total = B * world
    coll = describe_collective(world, dev, B, CLS)
    gather_ms = time_gather(world, dev, B, CLS)

    side = None
    side_coll = describe_collective(world, dev, args.bf16_batch, CLS) if world > 1 else None
    if args.config == 2 and args.bf16_batch > 0:
        side, wl16 = bf16_side_object(lib, args, world, rank,
                                      "BASELINE configs[2]" if world == 1 else "BASELINE configs[3]")
        if world == 1 and rank == 0:   # bf16 landmarks of the whole 512 batch against the fp32 HIP path, same crops
            a = wl16.model.forward_device(wl16.crops, "landmarks", n_points=0).cpu().numpy()
            b = wl.model.forward_device(wl16.crops, "landmarks", n_points=0).cpu().numpy()
            err = np.linalg.norm(a - b, axis=-1)
            side["landmark_nme_vs_fp32_hip"] = float(err.mean() / 256.0)
            side["max_coord_err_px"] = float(np.abs(a - b).max())
            side["parity_note"] = "all-pixel centroid, all %d faces of the timed batch (tests/test_gpu_baseline_configs.py " \
                                  "gates the same at batch 512 and checks the candidate path bit for bit)" % wl16.batch
        del wl16

    f32_big = None
    if args.config == 2 and world == 1 and args.f32_big_batch > 0:
        # the largest single-GPU configuration at the reference's own precision: does the 64-face roofline fraction hold
        # when fc6 runs eight generations of workgroups instead of two?  (enc2's 2 GiB input runs as two face slices)
        del wl
        torch.cuda.empty_cache()
        wlb = Workload("f32", args.f32_big_batch, rank, args.n_points, not args.no_align, seed=1)
        steps_b = max(3, args.steps // 4)
        dtb, fc6_b, layers_b, _ = measure(lib, wlb, steps_b, 2, 1, "fc6", args.settle_ms / 1e3)
        issued = fc6_issued_gflop(wlb.batch)
        f32_big = {"workload": "%d faces fp32 on one GPU, same step as the headline" % wlb.batch, "dtype": "f32",
                   "faces_per_s": wlb.batch * steps_b / dtb, "ms_per_step": 1e3 * dtb / steps_b, "steps": steps_b,
                   "roofline": mfma_roofline("fc6", "igemm_kernel<f32,MMAP=2,RELU,TWO> (fc6)", fc6_b, wlb.batch,
                                             PEAK_F32_TFLOPS, "traffic_latest.json",
                                             {"issued_tflops": issued / fc6_b, "frac_issued": issued / fc6_b / PEAK_F32_TFLOPS,
                                              "traffic": None}),
                   "forward": forward_summary(layers_b, wlb.batch, PEAK_F32_TFLOPS)}
        del wlb
        torch.cuda.empty_cache()
        wl = Workload(dtype, B, rank, args.n_points, not args.no_align, seed=1)   # (the parity block below runs on it)

    if rank == 0:
        value = total * args.steps / dt
        names = {2: "BASELINE configs[1]", 3: "BASELINE configs[2]", 4: "BASELINE configs[3]"}
        if bf16_head:
            roof = mfma_roofline("up3", "up3_cand8_kernel<bf16, 8 waves> (up3 + softmax + candidate keys)", roof_ms, B, peak,
                                 "traffic_bf16_latest.json")
        else:
            issued = fc6_issued_gflop(B)
            roof = mfma_roofline("fc6", "igemm_kernel<f32,MMAP=2,RELU,TWO> (fc6: 7x7x256->4096 on 8x8, M=64*B, K=12544; LDS-DMA "
                                        "operand ring, three-level accumulation)",
                                 roof_ms, B, peak, "traffic_latest.json",
                                 {"issued_tflops": issued / roof_ms, "frac_issued": issued / roof_ms / peak})
        rec = {
            "metric": "faces/sec (whole node), 256x256 batch inference: fused preprocess + FCN-8 forward + "
                      "softmax + top-%d landmark decode + similarity/alignment warp" % args.n_points,
            "value": value, "unit": "faces/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "per_rank_ms_per_step": per_rank, "gather_ms": gather_ms,
            "config": {"workload": "%s: batch=%d/GPU 256x256x3 uint8 crops, fcn_8(68) vanilla encoder %s, random-init "
                                   "weights (seed 2), decode top-%d thresh 0, align to 256x256%s"
                                   % (names[args.config], B, "fp32" if not bf16_head else "bf16 operands / fp32 accumulate",
                                      args.n_points, "" if not args.no_align else " (off)"),
                       "faces_per_gpu_per_step": B, "global_batch": total, "parallelism": "dp%d" % world,
                       "collective": coll, "settle_ms_before_timed_region": args.settle_ms},
            "roofline": roof,
            "forward": forward_summary(layer_avg, B, peak),
        }
        if args.cpu_faces > 0 and world == 1 and not bf16_head:   # CPU leg: rank 0 at N=1 only
            base, lm_cpu, crops_cpu = cpu_baseline(args.cpu_faces, args.n_points, seed=1)
            rec["cpu_baseline"] = base
            # the same crops on the GPU: landmark NME vs the oracle (NME := mean ||p - p_ref|| / 256)
            nb = len(crops_cpu)
            xd = torch.from_numpy(crops_cpu).to(dev)
            got = wl.model.forward_device(xd, "landmarks", n_points=args.n_points).cpu().numpy()
            lm64, gap = cpu_baseline.float64
            ref64, ref32 = lm64.reshape(nb, CLS, 2), lm_cpu.reshape(nb, CLS, 2)
            # a top-n centroid selects pixels: pairs whose n-th and (n+1)-th float64 probabilities lie within 2e-5
            # (relative) may legitimately select differently in any float32 evaluation -- counted, not priced
            det = np.ones((nb, CLS), bool) if gap is None else gap > 2e-5
            e64 = np.abs(got - ref64).max(-1)
            e32 = np.abs(ref32 - ref64).max(-1)
            rec["parity"] = {"landmark_nme_vs_oracle": float(np.linalg.norm(got - ref64, axis=-1)[det].mean() / 256.0),
                             "max_coord_err_px": float(e64[det].max()), "faces": nb, "pairs": int(det.size),
                             "undetermined_pairs": int((~det).sum()),
                             "max_coord_err_px_undetermined": float(e64[~det].max()) if (~det).any() else 0.0,
                             "float32_cpu_oracle_max_coord_err_px": float(e32[det].max()),
                             "note": "HIP fp32 landmarks (top-%d) of the CPU leg's crops against the FLOAT64 oracle decoded "
                                     "by the reference's arithmetic; north_star bar 1e-4 px.  Pairs whose n-th / (n+1)-th "
                                     "float64 probabilities lie within 2e-5 relative are counted as undetermined (any "
                                     "float32 evaluation may select another pixel there).  The float32 CPU oracle's own "
                                     "error on the same pairs rides along; the gate on all 64 faces of configs[1] is "
                                     "tests/test_gpu_baseline_configs.py" % args.n_points}
        if not args.no_hbm_kernels and world == 1:
            rec["hbm_kernels"] = hbm_kernels(lib, dev)
        if f32_big is not None:
            rec["f32_batch512"] = f32_big
        if side is not None:
            if world > 1:
                side["collective"] = side_coll
            rec["bf16_config3" if world == 1 else "config4"] = side
        print(json.dumps(rec))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
