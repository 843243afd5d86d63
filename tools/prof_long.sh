#!/bin/bash
# Profiles the long-read path (snk_long.hip) on the GPU box:  tools/prof_long.sh <tag> [L=1000] [pairs=1000000] [c2|c3] [pmc=1]
# kernel trace (+ three PMC passes when pmc=1) of tools/bench_long.py; per-kernel summary in gpurun_out/<tag>/summary.txt
set -u
TAG=${1:-long}; L=${2:-1000}; N=${3:-1000000}; WL=${4:-c2}; PMC=${5:-1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/tools/bench_long.py $L $N $WL"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o long -- $CMD > "$OUT/bench_trace.log" 2>&1
if [ "$PMC" = "1" ]; then
  i=0
  for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAVES SQ_BUSY_CYCLES" \
             "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $pmc -d "$OUT/pmc$i" -o long -- $CMD > "$OUT/bench_pmc$i.log" 2>&1
  done
fi
cd "$ROOT"
python - "$OUT" > "$OUT/summary.txt" <<'PY'
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
PY
grep "L=" "$OUT/bench_trace.log"
cat "$OUT/summary.txt"
