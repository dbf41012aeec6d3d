"""Turn rocprofv3 PMC passes of `bench.py` into profiles/ artefacts.

    python tools/pmc_report.py <dir with fetch/ write/ sq/ subdirs> <round tag>

Reads <dir>/{fetch,write,sq}/pmc_counter_collection.csv (+ kernel traces), maps kernels to layers by
name and grid size, and writes
  profiles/<tag>_pmc_per_kernel.csv   per-kernel means: duration, FETCH_SIZE, WRITE_SIZE, MFMA busy, clock
  profiles/traffic_latest.json        {layer: HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KB -> B)}
FETCH_SIZE is doubled as MI355X_MICROARCH.md "HBM" prescribes for wide (16 B/lane) coalesced reads on gfx950.
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(d):
    rows = list(csv.DictReader(open(os.path.join(d, "pmc_counter_collection.csv"))))
    kt = list(csv.DictReader(open(os.path.join(d, "pmc_kernel_trace.csv"))))
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        key = (r["Kernel_Name"], int(r["Grid_Size"]))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["_ns"].append(dur[r["Dispatch_Id"]])
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def layer_of(name, grid, batch=64):
    """Kernel (name, grid threads) -> layer label of the batch-64 fp32 bench step."""
    g = grid // 256
    if "enc1_kernel<" in name:
        return "enc1"
    if "decode_partial" in name:
        return "decode"
    if "warp_kernel" in name:
        return "warp"
    if "convt_kernel" in name:
        if name.rstrip(">(flm::ConvTArgs)").endswith(", 2") or ", 2>(" in name:
            return "up3_sub"
        return {162: "up5", 578: "up4"}.get(g, "up3" if g > 5000 else None)
    if "cand_merge" in name:
        return "decode"
    if "cand_tau" in name:
        return "tau"
    if "igemm_kernel<false, 2" in name or "igemm_kernel<0, 2" in name or "ILb0ELi2" in name:
        return "fc6"
    if "igemm_kernel" in name and ("false, 1" in name or "ILb0ELi1" in name):
        return {8192: "enc2", 4096: "enc3", 1024: "enc4", 256: "enc5"}.get(g)
    if "igemm_kernel" in name and ("false, 0, true" in name):
        return "fc7"
    return None


def main():
    base, tag = sys.argv[1], sys.argv[2]
    f = load(os.path.join(base, "fetch"))
    w = load(os.path.join(base, "write"))
    q = load(os.path.join(base, "sq"))
    out_rows = []
    traffic = {}
    for key in sorted(f, key=lambda k: -f[k]["_ns"]):
        name, grid = key
        if not any(t in name for t in ("flm::",)):
            continue
        fs = f[key].get("FETCH_SIZE", 0.0)
        ws = w.get(key, {}).get("WRITE_SIZE", 0.0)
        sq = q.get(key, {})
        cyc = sq.get("GRBM_GUI_ACTIVE", 0.0) / 8
        ns = sq.get("_ns", f[key]["_ns"])
        row = {"kernel": name[:90], "grid_threads": grid, "layer": layer_of(name, grid) or "",
               "avg_us": round(f[key]["_ns"] / 1e3, 1), "FETCH_SIZE_KB": round(fs, 1), "WRITE_SIZE_KB": round(ws, 1),
               "hbm_bytes_per_launch": int(2 * fs * 1024 + ws * 1024),
               "clock_GHz": round(cyc / ns, 3) if ns and cyc else "",
               "mfma_busy_frac": round(sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 3) if cyc else ""}
        out_rows.append(row)
        if row["layer"] and row["layer"] not in traffic:
            traffic[row["layer"]] = row["hbm_bytes_per_launch"]
    dst = os.path.join(ROOT, "profiles", "%s_pmc_per_kernel.csv" % tag)
    with open(dst, "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=list(out_rows[0].keys()))
        wr.writeheader()
        wr.writerows(out_rows)
    with open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w") as fh:
        json.dump(traffic, fh, indent=1, sort_keys=True)
    print("wrote", dst, "and traffic_latest.json:", traffic)


if __name__ == "__main__":
    main()
