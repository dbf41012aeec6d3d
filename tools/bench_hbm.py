#!/usr/bin/env python3
"""Developer tool: the HBM-bound kernels on their own (bench.py's hbm_kernels block): standalone decode and alignment warp."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import flm_amd  # noqa: E402,F401
from flm_amd import _lib  # noqa: E402

only = os.environ.get("ONLY")
r = bench.hbm_kernels(_lib.load(), torch.device("cuda", 0), only.split(",") if only else None)
for k, v in r.items():
    print("%-18s %.4f ms  %7.1f GB/s  %.3f of 8 TB/s" % (k, v["avg_launch_ms"], v["achieved"], v["frac"]))
print(json.dumps(r))
