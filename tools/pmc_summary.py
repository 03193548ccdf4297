"""Summarise a rocprofv3 --pmc --kernel-trace csv dir: per kernel name, average counters and derived ratios.
usage: python tools/pmc_summary.py <dir>"""
import csv, glob, os, sys, collections
d = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in sorted(rows.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    n = len(next(iter(cs.values())))
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    us = sum(dur[k]) / max(1, len(dur[k])) if k in dur else 0
    out = {"n": n, "us": round(us, 1)}
    wc = avg.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
                  "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM"):
            if c in avg: out[c.replace("SQ_", "") + "%wave"] = round(100 * avg[c] / wc, 1)
    if us and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        out["mfma_busy%"] = round(100 * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * us * 2400), 1)
    if us and "SQ_LDS_IDX_ACTIVE" in avg:
        out["lds_busy%"] = round(100 * avg["SQ_LDS_IDX_ACTIVE"] / (256 * us * 2400), 1)
    if "SQ_LDS_BANK_CONFLICT" in avg and avg.get("SQ_LDS_IDX_ACTIVE"):
        out["bank_conf%"] = round(100 * avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"], 1)
    for c in ("GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_WAVES"):
        if c in avg: out[c.replace("SQ_", "")] = int(avg[c])
    print(k.split("(ss::")[0].split("(float")[0][:120], out)
