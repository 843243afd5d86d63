"""Per-kernel summary of a tools/prof_long.sh output directory (kernel trace + PMC passes)."""
import csv, glob, os, sys
out = sys.argv[1]
def files(sub, pat): return glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
dur = {}
for f in files("trace", "*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if "snk_" in k: dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, d in dur.items():
    d = d[1:] if len(d) > 2 else d
    print("%-60s launches %d mean %.3f ms min %.3f" % (k[:60], len(d), sum(d) / len(d), min(d)))
cnt = {}
for p in sorted(glob.glob(os.path.join(out, "pmc*"))):
    for f in files(os.path.basename(p), "*counter_collection.csv"):
        acc = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            if "snk_" not in k: continue
            acc.setdefault((k, r["Counter_Name"]), {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[(k, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (k, c), v in acc.items():
            cnt.setdefault(k, {})[c] = sum(v.values()) / len(v)
for k, c in cnt.items():
    print(k[:60])
    print("   " + "  ".join("%s=%.4g" % (a, b) for a, b in sorted(c.items())))
    if "SQ_WAVE_CYCLES" in c:
        print("   of wave cycles: " + "  ".join("%s %.3f" % (a[3:].lower(), c[a] / c["SQ_WAVE_CYCLES"]) for a in sorted(c) if a.startswith("SQ_ACTIVE") or a.startswith("SQ_WAIT")))
    if "FETCH_SIZE" in c:
        print("   hbm bytes/launch (2*FETCH+WRITE, KB units): %.3f GB" % ((2 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0)) * 1024 / 1e9))
