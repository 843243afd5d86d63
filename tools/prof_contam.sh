#!/bin/bash
# rocprofv3 passes over the contaminant configuration of tools/bench_configs.py (gpurun, from the repo root):
#   tools/prof_contam.sh <tag> [config name filter]
set -u
TAG=${1:-contam}
FILTER=${2:-contam1/2 + global}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o c -- python $ROOT/tools/bench_configs.py "$FILTER" > "$OUT/trace.log" 2>&1
i=0
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $pmc -d "$OUT/pmc$i" -o c -- python $ROOT/tools/bench_configs.py "$FILTER" > "$OUT/pmc$i.log" 2>&1
done
cd "$ROOT"
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs cat | head -8
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/pmc*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            print(k, {c: round(x / cnt[(k, c)]) for c, x in v.items()})
PY
