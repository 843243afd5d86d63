"""Summarises the rocprofv3 outputs of tools/profile.sh: mean duration of the hot kernel from the
kernel trace and mean per-launch counter values from the PMC passes (hot kernel only)."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
HOT = "snk_tiled_kernel"
res = {"kernel": None, "trace": {}, "counters": {}}


def files(sub, pat):
    return glob.glob(os.path.join(out, sub, "**", pat), recursive=True)


for f in files("trace", "*kernel_trace.csv"):
    d = []
    for r in csv.DictReader(open(f)):
        if HOT in r["Kernel_Name"]:
            res["kernel"] = r["Kernel_Name"].split("(")[0]
            d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    if d:
        d = d[2:] if len(d) > 4 else d          # skip warm-up launches
        res["trace"] = {"launches": len(d), "mean_ms": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d)}
for p in sorted(glob.glob(os.path.join(out, "pmc*"))):
    for f in files(os.path.basename(p), "*counter_collection.csv"):
        acc = {}
        for r in csv.DictReader(open(f)):
            if HOT not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            vals = list(v.values())
            res["counters"][k] = {"launches": len(vals), "mean_per_launch": sum(vals) / len(vals)}
c = {k: v["mean_per_launch"] for k, v in res["counters"].items()}
reads = 20e6
der = {}
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_FLAT", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"):
    if k in c:
        der[k.replace("SQ_INSTS_", "").lower() + "_per_read"] = c[k] / reads
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # guide: FETCH_SIZE reports half the bytes of wide coalesced streaming reads on gfx950 (KB units)
    der["hbm_bytes_per_launch"] = 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
    der["fetch_kb_raw"] = c["FETCH_SIZE"]
    der["write_kb_raw"] = c["WRITE_SIZE"]
if "SQ_WAVE_CYCLES" in c:
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_SCA",
              "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
        if k in c:
            der[k.lower() + "_frac_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
res["derived"] = der
print(json.dumps(res, indent=1))
