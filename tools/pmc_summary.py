"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean over dispatches)."""
import csv, sys, collections
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    key = (k, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, c in agg.items():
    name = key[0].replace("void flm::", "").replace("flm::", "")[:60]
    m = {k: sum(v) / len(v) for k, v in c.items()}
    n = len(next(iter(c.values())))
    line = "%-60s grid %-9s n=%-3d" % (name, key[1], n)
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc:
        line += " mfma_busy/busy %.2f" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(m.get("SQ_BUSY_CYCLES", 1), 1))
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in m:
                line += " %s %.2f" % (k[3:], m[k] / wc)
        if "SQ_LDS_BANK_CONFLICT" in m:
            line += " ldsconf/wavecyc %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / wc)
    for k in m:
        if k.startswith("GRBM") or k in ("FETCH_SIZE", "WRITE_SIZE") or k.startswith("TCC"):
            line += " %s %.4g" % (k, m[k])
    print(line)
    if "-v" in sys.argv:
        print("    ", {k: "%.4g" % v for k, v in m.items()})
