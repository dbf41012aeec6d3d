"""Per-kernel means of the bf16 batch-512 PMC passes (tools/prof_bf16.py under rocprofv3) -> profiles/<tag>_bf16_pmc_per_kernel.csv

    python tools/pmc_bf16_report.py <dir with sq/ fetch/ write/ subdirs> <round tag>
"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_report import load, ROOT

base, tag = sys.argv[1], sys.argv[2]
q = load(os.path.join(base, "sq"))
f = load(os.path.join(base, "fetch")) if os.path.isdir(os.path.join(base, "fetch")) else {}
w = load(os.path.join(base, "write")) if os.path.isdir(os.path.join(base, "write")) else {}
rows = []
for key in sorted(q, key=lambda k: -q[k]["_ns"]):
    name, grid = key
    if "flm::" not in name or q[key]["_ns"] < 20e3:
        continue
    cyc = q[key].get("GRBM_GUI_ACTIVE", 0.0) / 8
    fs = f.get(key, {}).get("FETCH_SIZE", 0.0)
    ws = w.get(key, {}).get("WRITE_SIZE", 0.0)
    rows.append({"kernel": name[:100], "grid_threads": grid, "avg_us": round(q[key]["_ns"] / 1e3, 1),
                 "clock_GHz": round(cyc / q[key]["_ns"], 3) if cyc else "",
                 "mfma_busy_frac": round(q[key].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 3) if cyc else "",
                 "wait_inst_frac": round(q[key].get("SQ_WAIT_INST_ANY", 0.0) / max(q[key].get("SQ_WAVE_CYCLES", 1.0), 1.0), 3),
                 "hbm_bytes_per_launch": int(2 * fs * 1024 + ws * 1024) if (fs or ws) else ""})
dst = os.path.join(ROOT, "profiles", "%s_bf16_pmc_per_kernel.csv" % tag)
with open(dst, "w", newline="") as fh:
    wr = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    wr.writeheader()
    wr.writerows(rows)
print("wrote", dst, len(rows), "kernels")
