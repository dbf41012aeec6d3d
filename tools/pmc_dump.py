#!/usr/bin/env python3
"""Developer tool: mean PMC counter values per kernel from rocprofv3 --pmc passes.

    python tools/pmc_dump.py <pattern in kernel name> <dir> [<dir> ...]     (each dir holds pmc_counter_collection.csv)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_report import load  # noqa: E402

pat = sys.argv[1]
for d in sys.argv[2:]:
    q = load(d)
    for (name, grid), c in sorted(q.items(), key=lambda kv: -kv[1]["_ns"]):
        if pat not in name:
            continue
        print("%s grid %d  %.1f us" % (name[:90], grid, c["_ns"] / 1e3))
        for k, v in sorted(c.items()):
            if k != "_ns":
                print("    %-32s %.4g" % (k, v))
