"""Developer tool: sampling-launch phases per tile (flm_forward_opts.candidate_sub_phases) vs time, list fill and the
fallback flag, headline batch (fp32, 64) and bf16 batch 512."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
from flm_amd import _lib
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights

lib = _lib.load()
w = synth_fcn8_weights(68, 2)
for dtype, B in (("f32", 64), ("bf16", 512)):
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dtype)
    model.load_weights(w)
    x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
    for npts in [int(v) for v in os.environ.get("NPTS", "4,25").split(",")]:
        for sub in [int(v) for v in os.environ.get("SUBS", "4,3,2,1").split(",")]:
            opts = dict(candidate_sub_phases=sub)
            ws = model.new_workspace(B, "landmarks", npts, opts)
            for _ in range(3):
                model.forward_device(x, "landmarks", n_points=npts, workspace=ws, opts=opts)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                model.forward_device(x, "landmarks", n_points=npts, workspace=ws, opts=opts)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            fo = model._opts(opts)
            import ctypes as C
            off = lambda name: lib.flm_fcn8_workspace_offset_opts(name, B, 256, 256, 68, model._dt, _lib.OUT_LANDMARKS,
                                                                  _lib.DECODE_TOPN, npts, C.byref(fo))
            cap = off(b"cand_cap")
            cnt = ws[off(b"cand_cnt"):off(b"cand_cnt") + 4 * (B + 1)].view(torch.int32).cpu().numpy()
            print("%s n_points %2d sub %d: %.3f ms  keys/face min %d max %d (cap %d)  fallback %d" %
                  (dtype, npts, sub, 1e3 * dt, cnt[:B].min(), cnt[:B].max(), cap, cnt[B]), flush=True)
