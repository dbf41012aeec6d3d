"""Developer tool: wall-clock latency of one small batch (pipelined launches and fully synchronous calls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flm_amd
from flm_amd.networks import LANDMARKS_MODELS
from flm_amd.weights import synth_fcn8_weights
w = synth_fcn8_weights(68, 2)
for dtype in ("f32", "bf16"):
    m = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype=dtype)
    m.load_weights(w)
    for B in (1, 8):
        x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
        for _ in range(10):
            m.forward_device(x, "landmarks", n_points=4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            m.forward_device(x, "landmarks", n_points=4)
        torch.cuda.synchronize()
        pipelined = (time.perf_counter() - t0) / 200
        t0 = time.perf_counter()
        for _ in range(100):
            m.forward_device(x, "landmarks", n_points=4)
            torch.cuda.synchronize()
        sync = (time.perf_counter() - t0) / 100
        print("%s batch %d: %.3f ms per call back to back, %.3f ms per call with a sync after each" % (dtype, B, 1e3 * pipelined, 1e3 * sync))
        # the same plus similarity + warp: eager launches vs one HIP graph (graphs.CapturedPipeline)
        from flm_amd import alignment, graphs
        pipe = graphs.CapturedPipeline(m, B, n_points=4)
        def eager():
            lm = m.forward_device(x, "landmarks", n_points=4)
            return alignment.align_device(x, lm, pipe.template, 256, 256, pipe.scale)
        res = {}
        for name, fn in (("eager", eager), ("graph", lambda: pipe(x))):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                fn()
            torch.cuda.synchronize()
            back = (time.perf_counter() - t0) / 200
            t0 = time.perf_counter()
            for _ in range(100):
                fn()
                torch.cuda.synchronize()
            res[name] = (back, (time.perf_counter() - t0) / 100)
        print("%s batch %d landmarks + align: eager %.3f / %.3f ms, graph %.3f / %.3f ms (back to back / synced)" %
              (dtype, B, 1e3 * res["eager"][0], 1e3 * res["eager"][1], 1e3 * res["graph"][0], 1e3 * res["graph"][1]), flush=True)
        del pipe
