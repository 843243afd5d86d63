#!/bin/bash
# rocprofv3 kernel statistics of one CLI run (.gz -> .gz, C2 parameters): tools/prof_cli.sh <pairs> <tag>
N=${1:-4000000}; TAG=${2:-cli}
ROOT=$(pwd); TMP=$(mktemp -d /dev/shm/snkp_XXXX); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
python - "$TMP" "$N" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_e2e
bench_e2e.make_inputs(sys.argv[1], int(sys.argv[2]), ["gz"])
PY
export TMPDIR=/tmp; cd /tmp
A="-f AAGTCGGAGGCCAAGCGGTCTTAGGAAGACAA -r AAGTCGGATCGTAGCCATGTCGTTCTGTGAGCCAAGGAGTTG -J -l 10 -q 0.1 -T 16"
SNK_CLEAN_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o cli -- $ROOT/soapnuke_amd/SOAPnuke filter -1 $TMP/r1.fq.gz -2 $TMP/r2.fq.gz -C c1.fq.gz -D c2.fq.gz -o $TMP/out $A > $OUT/run.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cut -c1-60,200- $OUT/kernel_stats.csv | head -30
python - $OUT/kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %6s  avg %10.1f us  total %8.1f ms  %5s %%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
rm -rf $TMP
