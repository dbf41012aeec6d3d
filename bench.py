#!/usr/bin/env python3
"""Benchmark of the landmark hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic 256x256 uint8 BGR face crops
already resident in HBM (BASELINE.json configs[1]: batch 64, fp32):
    fused preprocess + FCN-8 forward -> softmax probabilities [64, 264*264, 68]
    -> top-n landmark decode (n=4, thresh=0: the reference's as-shipped decode)
    -> similarity fit + alignment warp to 256x256
    -> (N > 1) RCCL all-gather of the landmark tensors.
Ranks are weak-scaled: every rank runs its own batch (no data-path collective besides the
landmark gather).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import concurrent.futures
import ctypes as C
import json
import os
import sys
import time

# the pool's host driver only supports dmabuf IPC: RCCL fails with hipIpcGetMemHandle errors without this (set before HIP starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GFLOP_PER_FACE = 17.844          # 2*MAC over every Conv2D/Conv2DTranspose, dense
FC6_GFLOP_PER_FACE = 6.5767      # fc6 7x7x256x4096 on 8x8, dense count (includes zero-padded taps)
PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = FP32 vector peak
PEAK_BF16_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA


def fc6_issued_gflop(batch):
    """MACs fc6 actually issues at this batch: position-major 128-row tiles skip the filter taps that see
    only zero padding for the whole tile (csrc/flm_igemm_f32.hip); returned as 2*MAC in GFLOP."""
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


def load_traffic(layer):
    """HBM bytes per launch of a layer's kernel from the committed PMC summary (None when absent):
    profiles/traffic_latest.json = {layer: FETCH_SIZE*2 + WRITE_SIZE in bytes}, see profiles/README.md."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            return json.load(f).get(layer)
    except (OSError, ValueError):
        return None


# decode side of the step: the landmark selection ("up3_sub" + "tau" = its threshold pass over a 1/16 sample of the
# map) and the gated fallback launches; everything else is the FCN forward whose FLOPs SURVEY 8(d) counts
DECODE_KEYS = ("decode", "up3_sub", "tau", "decode_fallback", "up3_fallback")


RECORDS_PER_STEP = 24   # launches one step brackets at most
LAYER_PASS_STEPS = 3    # steps of the separate per-layer timing pass


def read_profile(lib):
    """Per-layer mean duration (ms) of the launches recorded since the last reset."""
    import numpy as np
    layer_ms = {}
    name = C.create_string_buffer(32)
    ms = C.c_float()
    i = 0
    while lib.flm_profile_read(i, name, 32, C.byref(ms)) == 0:
        layer_ms.setdefault(name.value.decode(), []).append(ms.value)
        i += 1
    return {k: float(np.mean(v)) for k, v in layer_ms.items()}


def bf16_config3(lib, dev, batch, steps, warmup, n_points, fp32_model):
    """BASELINE configs[2]: batch-512 bf16 conv stack (bf16 operands, fp32 accumulate) on one GPU: same step
    as the headline (forward + softmax + decode + align); reported beside it, never as `value` (the
    reference computes in fp32).  NME: bf16 landmarks vs the fp32 HIP path on the same crops, all-pixel
    centroid (a top-n selection amplifies bf16 noise into pixel jumps on random-weight heatmaps)."""
    import numpy as np
    import torch
    from flm_amd import _lib, alignment
    from flm_amd.networks import LANDMARKS_MODELS
    from flm_amd.weights import synth_fcn8_weights
    H = W = 256
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=H, input_width=W, dtype="bf16")
    model.load_weights(synth_fcn8_weights(68, seed=2))
    crops = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (batch, H, W, 3), dtype=np.uint8)).to(dev)
    tmpl = torch.from_numpy(alignment.canonical_template(68, H, W)).to(dev)
    scale = (W / model.output_width, H / model.output_height)

    def step():
        lm = model.forward_device(crops, "landmarks", n_points=n_points, thresh=0.0)
        alignment.align_device(crops, lm, tmpl, H, W, scale)
        return lm

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # per-layer durations: a separate pass with every launch bracketed by HIP events (outside the timed region)
    _lib.check(lib.flm_profile_enable(LAYER_PASS_STEPS * RECORDS_PER_STEP + 64), "flm_profile_enable")
    lib.flm_profile_filter(None)
    lib.flm_profile_reset()
    for _ in range(LAYER_PASS_STEPS):
        step()
    torch.cuda.synchronize()
    layers = read_profile(lib)
    lib.flm_profile_disable()
    fwd_ms = sum(v for k, v in layers.items() if k not in DECODE_KEYS)
    nb = min(64, batch)
    a = model.forward_device(crops[:nb].contiguous(), "landmarks", n_points=0).cpu().numpy()
    b = fp32_model.forward_device(crops[:nb].contiguous(), "landmarks", n_points=0).cpu().numpy()
    err = np.linalg.norm(a - b, axis=-1)
    return {"workload": "BASELINE configs[2]: batch=%d 256x256 crops, bf16 operands / fp32 accumulate, same step "
                        "as the headline" % batch,
            "dtype": "bf16", "faces_per_s": batch * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "forward_ms": fwd_ms, "forward_tflops": GFLOP_PER_FACE * batch / fwd_ms,
            "frac_of_bf16_mfma_peak": GFLOP_PER_FACE * batch / fwd_ms / PEAK_BF16_TFLOPS,
            "layer_ms": layers,
            "landmark_nme_vs_fp32_hip": float(err.mean() / 256.0), "max_coord_err_px": float(np.abs(a - b).max())}


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


def cpu_baseline(n_faces, n_points, seed):
    """The oracle (kind "port": the build's CPU restatement of prediction.py's path) timed on the
    host cores: preprocess + FCN-8 forward + softmax + top-n decode for `n_faces` crops."""
    import numpy as np
    import torch
    from flm_amd.weights import synth_fcn8_weights
    from oracle import decode_ref, fcn_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    params = synth_fcn8_weights(68, seed=2)
    rng = np.random.default_rng(seed)
    crops = rng.integers(0, 256, (n_faces, 256, 256, 3), dtype=np.uint8)

    def decode_face(hm_one):  # the reference's per-landmark argsort loop (utils/metrics.py:66-77), one face
        with np.errstate(all="ignore"):
            return decode_ref.transfer_target_ref(hm_one, 0, n_points)

    def run(c):
        x = np.stack([fcn_ref.get_image_array_ref(i) for i in c])
        pr = fcn_ref.fcn8_predict_ref(x, params).reshape(len(c), 264, 264, 68)
        # faces are independent: one decode per worker thread (numpy's sort releases the GIL)
        with concurrent.futures.ThreadPoolExecutor(max_workers=cores) as ex:
            return np.concatenate(list(ex.map(decode_face, [pr[i:i + 1] for i in range(len(c))])))

    run(crops[:1])  # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    lm = run(crops)
    dt = time.perf_counter() - t0
    return {"value": n_faces / dt, "unit": "faces/s", "cores": cores, "kind": "port",
            "sample": "%d synthetic 256x256 crops through oracle/ (torch-CPU fp32 forward on %d threads + numpy "
                      "top-%d decode, one face per thread), %.1f s" % (n_faces, cores, n_points, dt)}, lm, crops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="faces per GPU per step")
    ap.add_argument("--n-points", type=int, default=4)
    ap.add_argument("--cpu-faces", type=int, default=64, help="sample size of the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-align", action="store_true")
    ap.add_argument("--bf16-batch", type=int, default=512,
                    help="also time BASELINE configs[2] (bf16 operands, this many faces per step) on rank 0 "
                         "and report it as a side object; 0 = skip")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import flm_amd  # noqa: F401
    from flm_amd import _lib, alignment, distributed
    from flm_amd.networks import LANDMARKS_MODELS
    from flm_amd.weights import synth_fcn8_weights

    rank, local_rank, world = distributed.env_world()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank if world > 1 else 0)
    if world > 1:
        distributed.init_process_group("nccl")
    dev = torch.device("cuda", torch.cuda.current_device())
    lib = _lib.load()

    B, H, W, CLS = args.batch, 256, 256, 68
    model = LANDMARKS_MODELS["fcn_8"](CLS, input_height=H, input_width=W)
    model.load_weights(synth_fcn8_weights(CLS, seed=2))
    rng = np.random.default_rng(1 + rank)
    crops = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).to(dev)
    tmpl = torch.from_numpy(alignment.canonical_template(CLS, H, W)).to(dev)
    scale = (model.input_width / model.output_width, model.input_height / model.output_height)
    total = B * world

    def step():
        lm = model.forward_device(crops, "landmarks", n_points=args.n_points, thresh=0.0)
        if not args.no_align:
            aligned, m = alignment.align_device(crops, lm, tmpl, H, W, scale)
        else:
            aligned = None
        full = distributed.all_gather_landmarks(lm, total) if world > 1 else lm
        return full, aligned

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # Timed region: only the roofline kernel (fc6) is bracketed by HIP events on the launch stream -- an event pair
    # around each of the ~18 launches of a step costs 1.7 % of the step.  The per-layer table comes from a separate
    # pass right after it.
    _lib.check(lib.flm_profile_enable(max(args.steps, LAYER_PASS_STEPS) * RECORDS_PER_STEP + 64), "flm_profile_enable")
    _lib.check(lib.flm_profile_filter(b"fc6"), "flm_profile_filter")
    lib.flm_profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full, aligned = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- fc6 launch durations recorded by HIP events inside the timed region; then the per-layer pass ----
    fc6_timed = read_profile(lib).get("fc6", float("nan"))
    _lib.check(lib.flm_profile_filter(None), "flm_profile_filter")
    lib.flm_profile_reset()
    for _ in range(LAYER_PASS_STEPS):
        step()
    fence()
    layer_avg = read_profile(lib)
    lib.flm_profile_disable()

    if rank == 0:
        value = total * args.steps / dt
        fwd_keys = [k for k in layer_avg if k not in DECODE_KEYS]
        fwd_ms = sum(layer_avg[k] for k in fwd_keys)
        fc6_ms = fc6_timed
        fc6_tflops = FC6_GFLOP_PER_FACE * B / fc6_ms  # GFLOP / ms = TFLOP/s
        fwd_tflops = GFLOP_PER_FACE * B / fwd_ms
        rec = {
            "metric": "faces/sec (whole node), 256x256 batch inference: fused preprocess + FCN-8 forward + "
                      "softmax + top-%d landmark decode + similarity/alignment warp" % args.n_points,
            "value": value, "unit": "faces/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch=%d/GPU 256x256x3 uint8 crops, fcn_8(68) vanilla "
                                   "encoder fp32, random-init weights (seed 2), decode top-%d thresh 0, "
                                   "align to 256x256%s" % (B, args.n_points, "" if not args.no_align else " (off)"),
                       "faces_per_gpu_per_step": B, "parallelism": "dp%d" % world,
                       "collective": "all_gather landmarks [B,68,2] f64" if world > 1 else "none"},
            "roofline": {"bound": "mfma",
                         "kernel": "igemm_f32_kernel<MMAP=2,RELU> (fc6: 7x7x256->4096 on 8x8, M=64*B, K=12544)",
                         "achieved": fc6_tflops, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": fc6_tflops / PEAK_F32_TFLOPS, "traffic": load_traffic("fc6"),
                         "flop_per_launch": FC6_GFLOP_PER_FACE * 1e9 * B, "avg_launch_ms": fc6_ms,
                         "issued_tflops": fc6_issued_gflop(B) / fc6_ms,
                         "frac_issued": fc6_issued_gflop(B) / fc6_ms / PEAK_F32_TFLOPS,
                         "note": "achieved counts the dense 2*MAC of SURVEY 8(d) (zero-padded taps included); the "
                                 "kernel skips taps that only see padding, so fewer MFMAs are issued: "
                                 "issued_tflops / frac_issued price the matrix pipe itself",
                         "traffic_source": "profiles/ (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"},
            "forward": {"gflop_per_face": GFLOP_PER_FACE, "ms": fwd_ms, "tflops": fwd_tflops,
                        "frac_of_f32_mfma_peak": fwd_tflops / PEAK_F32_TFLOPS,
                        "faces_per_s_forward_only": 1e3 * B / fwd_ms, "layer_ms": layer_avg,
                        "layer_ms_source": "separate pass of %d steps right after the timed region, every launch "
                                           "bracketed by HIP events (the timed region brackets fc6 only)"
                                           % LAYER_PASS_STEPS},
        }
        if args.cpu_faces > 0 and world == 1:   # CPU leg: rank 0 at N=1 only
            base, lm_cpu, crops_cpu = cpu_baseline(args.cpu_faces, args.n_points, seed=1)
            rec["cpu_baseline"] = base
            # the same crops on the GPU: landmark NME vs the oracle (NME := mean ||p - p_ref|| / 256)
            nb = min(args.cpu_faces, B)
            xd = torch.from_numpy(crops_cpu[:nb]).to(dev)
            got = model.forward_device(xd, "landmarks", n_points=args.n_points).cpu().numpy()
            ref = lm_cpu[:nb].reshape(nb, CLS, 2)
            err = np.linalg.norm(got - ref, axis=-1)
            rec["parity"] = {"landmark_nme_vs_oracle": float(err.mean() / 256.0),
                             "max_coord_err_px": float(np.abs(got - ref).max()), "faces": nb,
                             "note": "against the float32 oracle; a top-4 centroid divides by the sum of four ~1e-2 "
                                     "probabilities, so two float32 forwards with different summation orders differ "
                                     "by up to ~1e-4 px there -- tests/test_gpu_forward.py gates both against the "
                                     "oracle's float64 evaluation"}
        if args.bf16_batch > 0 and world == 1:
            rec["bf16_config3"] = bf16_config3(lib, dev, args.bf16_batch, max(3, args.steps // 2), 2, args.n_points,
                                               model)
        print(json.dumps(rec))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
