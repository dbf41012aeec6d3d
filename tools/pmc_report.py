#!/usr/bin/env python3
"""Turn the passes of tools/final_profile.sh into profiles/ artefacts.

    python tools/pmc_report.py gpurun_out/final_r02 r02

Per mode (f32 headline, bf16 configuration, HBM-bound kernels) reads <dir>/<mode>/{fetch,write,sq}/**/pmc_counter_collection.csv
and writes
  profiles/<tag>_pmc_per_kernel.csv, profiles/<tag>_bf16_pmc_per_kernel.csv, profiles/<tag>_hbm_pmc_per_kernel.csv
      per (kernel, grid): mean duration, FETCH_SIZE, WRITE_SIZE, HBM-side bytes per launch, clock, MFMA-busy, wait fractions
  profiles/traffic_latest.json, profiles/traffic_bf16_latest.json   {layer: HBM-side bytes per launch}, read by bench.py
HBM-side bytes = 2 * FETCH_SIZE + WRITE_SIZE (KB -> B): FETCH_SIZE doubled as MI355X_MICROARCH.md "HBM" prescribes for wide
(16 B/lane) coalesced reads on gfx950.  Also copies the kernel-trace stats and the bench lines.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(d):
    """Per (kernel name, grid size): means of every counter of one rocprofv3 --pmc pass, plus the mean duration (_ns)."""
    rows = list(csv.DictReader(open(os.path.join(d, "pmc_counter_collection.csv"))))
    kt = list(csv.DictReader(open(os.path.join(d, "pmc_kernel_trace.csv"))))
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        key = (r["Kernel_Name"], int(r["Grid_Size"]))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["_ns"].append(dur[r["Dispatch_Id"]])
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def find(base):
    hits = glob.glob(os.path.join(base, "**", "pmc_counter_collection.csv"), recursive=True)
    return os.path.dirname(hits[0]) if hits else None


def layer_f32(name, grid):
    g = grid // 256
    if "enc1_kernel<" in name:
        return "enc1"
    if "cand_merge" in name:
        return "decode"
    if "cand_tau" in name:
        return "tau"
    if "flm::warp_" in name:
        return "warp"
    if "similarity" in name:
        return "similarity"
    if "up3_cand8_kernel" in name:
        return "up3"
    if "convt_kernel" in name:
        if ", 2, false>" in name or ", 2>(" in name:
            return "up3_sub"
        if ", 1, true>" in name:
            return "up3"
        return "up5" if g < 300 else ("up4" if g < 1500 else None)
    if "igemm_kernel<false, 2" in name:
        return "fc6"
    if "igemm_kernel<false, 1" in name:
        return {8192: "enc2", 4096: "enc3", 1024: "enc4", 256: "enc5"}.get(g)
    if "igemm_kernel<false, 0, true" in name:
        return "fc7"
    return None


def layer_bf16(name, grid):
    g = grid // 256
    if "enc1_bf16_kernel" in name:
        return "enc1"
    if "conv3_halo" in name:
        return "enc2"
    if "igemm_bf16_big_kernel<2" in name:
        return "fc6"
    if "igemm_bf16_big_kernel<0" in name:
        return "fc7"
    if "igemm_bf16_big_kernel<1" in name:
        return None          # enc3 / enc4 / enc5 share the instantiation: told apart by duration below
    if "up3_cand8_kernel" in name:
        return "up3"
    if "seg_fused_bf16_kernel" in name:
        return "seg_fused"
    if "convt_kernel" in name and (", 2, false>" in name):
        return "up3_sub"
    if "cand_merge" in name:
        return "decode"
    if "cand_tau" in name:
        return "tau"
    if "flm::warp_" in name:
        return "warp"
    return None


def layer_hbm(name, grid):
    if "decode_partial" in name:
        # batch 64 and batch 512 launch the same grid (about 1536 workgroups either way): the per-(kernel, grid) means
        # of this table mix the two, so no traffic entry is derived for the standalone decode (round 1 measured it on
        # its own: 1.218 GB per batch-64 launch against 1.213 GB algorithmic)
        return None
    if "flm::warp_" in name:
        # warp_u8_rows_kernel<2> runs a thread per 2 output pixels (warp_u8_kernel<4>: per 4): 64 faces <= 2,097,152 threads
        return "warp_b64" if grid <= 64 * 256 * 256 // 2 else "warp_b512"
    return None


def standalone_decode_traffic(base):
    """tools/dec_traffic.sh: the standalone decode alone, one batch size per profiled run ->
    {decode_standalone_b<N>: HBM-side bytes per decode (all its launches)}."""
    out = {}
    for b in (64, 512):
        f, w = find(os.path.join(base, "b%d" % b, "FETCH_SIZE")), find(os.path.join(base, "b%d" % b, "WRITE_SIZE"))
        if not f or not w:
            continue
        fl, wl = load(f), load(w)
        tot = 0.0
        for key, v in fl.items():
            if "flm::decode_" in key[0]:
                tot += 2 * v.get("FETCH_SIZE", 0.0) * 1024 + wl.get(key, {}).get("WRITE_SIZE", 0.0) * 1024
        if tot:
            out["decode_standalone_b%d" % b] = int(tot)
    return out


def report(base, tag, suffix, layer_of, traffic_file, merge=False):
    dirs = {k: find(os.path.join(base, k)) for k in ("fetch", "write", "sq")}
    if not all(dirs.values()):
        print("skip", base, dirs)
        return
    f, w, q = (load(dirs[k]) for k in ("fetch", "write", "sq"))
    rows, traffic = [], {}
    for key in sorted(f, key=lambda k: -f[k]["_ns"]):
        name, grid = key
        if "flm::" not in name:
            continue
        fs, ws = f[key].get("FETCH_SIZE", 0.0), w.get(key, {}).get("WRITE_SIZE", 0.0)
        sq = q.get(key, {})
        cyc = sq.get("GRBM_GUI_ACTIVE", 0.0) / 8
        ns = sq.get("_ns", f[key]["_ns"])
        wave = max(sq.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        row = {"kernel": name[:100], "grid_threads": grid, "layer": layer_of(name, grid) or "",
               "avg_us": round(f[key]["_ns"] / 1e3, 1), "FETCH_SIZE_KB": round(fs, 1), "WRITE_SIZE_KB": round(ws, 1),
               "hbm_bytes_per_launch": int(2 * fs * 1024 + ws * 1024),
               "hbm_GBps": round((2 * fs * 1024 + ws * 1024) / f[key]["_ns"], 1),
               "clock_GHz": round(cyc / ns, 3) if ns and cyc else "",
               "mfma_busy_frac": round(sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 3) if cyc else "",
               "wait_any_frac": round(sq.get("SQ_WAIT_ANY", 0.0) / wave, 3),
               "wait_inst_frac": round(sq.get("SQ_WAIT_INST_ANY", 0.0) / wave, 3),
               "active_inst_frac": round(sq.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 3)}
        rows.append(row)
        if row["layer"] and row["layer"] not in traffic:
            traffic[row["layer"]] = row["hbm_bytes_per_launch"]
    if not rows:
        return
    dst = os.path.join(ROOT, "profiles", "%s%s_pmc_per_kernel.csv" % (tag, suffix))
    with open(dst, "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        wr.writerows(rows)
    tf = os.path.join(ROOT, "profiles", traffic_file)
    old = {}
    if merge and os.path.exists(tf):
        old = json.load(open(tf))
    old.update(traffic)
    with open(tf, "w") as fh:
        json.dump(old, fh, indent=1, sort_keys=True)
    print("wrote", dst, "and", traffic_file, traffic)


def main():
    base, tag = sys.argv[1], sys.argv[2]
    report(os.path.join(base, "f32"), tag, "", layer_f32, "traffic_latest.json")
    report(os.path.join(base, "hbm"), tag, "_hbm", layer_hbm, "traffic_latest.json", merge=True)
    report(os.path.join(base, "bf16"), tag, "_bf16", layer_bf16, "traffic_bf16_latest.json")
    dec = standalone_decode_traffic(os.path.join(os.path.dirname(os.path.abspath(base)), "dec_traffic"))
    if dec:
        tf = os.path.join(ROOT, "profiles", "traffic_latest.json")
        cur = json.load(open(tf))
        cur.update(dec)
        with open(tf, "w") as fh:
            json.dump(cur, fh, indent=1, sort_keys=True)
        print("standalone decode", dec)
    for sub, out in (("stats", "%s_bench_kernel_stats.csv"), ("stats_bf16", "%s_bf16_kernel_stats.csv"),
                     ("stats_hbm", "%s_hbm_kernel_stats.csv")):
        hits = glob.glob(os.path.join(base, sub, "**", "*kernel_stats.csv"), recursive=True)
        if hits:
            shutil.copy(hits[0], os.path.join(ROOT, "profiles", out % tag))
    for log, out in (("bench.log", "%s_bench_line.json"), ("stats.log", "%s_bench_line_under_rocprof.json"),
                     ("stats_bf16.log", "%s_bf16_bench_line_under_rocprof.json")):
        pth = os.path.join(base, log)
        if os.path.exists(pth):
            for line in open(pth):
                if line.startswith("{"):
                    with open(os.path.join(ROOT, "profiles", out % tag), "w") as fh:
                        fh.write(json.dumps(json.loads(line), indent=1) + "\n")
    pk = os.path.join(base, "mfma_peak.txt")
    if os.path.exists(pk):
        shutil.copy(pk, os.path.join(ROOT, "profiles", "%s_mfma_peak.txt" % tag))


if __name__ == "__main__":
    main()
