#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/micro/fetch_calib per kernel, next to the bytes it really reads (GPU box; run through gpurun)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/fetch_calib; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/p -o f -- $ROOT/tools/micro/fetch_calib > $OUT/log 2>&1
cat $OUT/log | grep known
python - $OUT/p <<'PY'
import csv, glob, os, sys
known = {"rd16": 2 << 30, "rd4": 2 << 30, "rd1(": 2 << 30, "rd1_rows": ((2 << 30) // 160) * 150, "rd_dma": 2 << 30}
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"]] = acc.get(r["Kernel_Name"], 0.0) + float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    kb = [b for n, b in known.items() if n.rstrip("(") in k and (n != "rd1(" or "rows" not in k)]
    if not kb:
        continue
    b = kb[0]
    print(f"{k[:40]:40s} FETCH_SIZE {v:14.0f} KB-units  known {b / 1024:14.0f} KB   bytes per counted KB-unit: {b / (v * 1024):.3f} x")
PY
