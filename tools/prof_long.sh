#!/bin/bash
# Profiles the long-read path (snk_long.hip) on the GPU box:  tools/prof_long.sh <tag> [L=1000] [pairs=1000000] [c2|c3|contam] [pmc=1] [kernel=0]
# kernel trace (+ three PMC passes when pmc=1) of tools/bench_long.py; per-kernel summary in gpurun_out/<tag>/summary.txt
set -u
TAG=${1:-long}; L=${2:-1000}; N=${3:-1000000}; WL=${4:-c2}; PMC=${5:-1}; KERN=${6:-0}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/tools/bench_long.py $L $N $WL $KERN"
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o long -- $CMD > "$OUT/bench_trace.log" 2>&1
if [ "$PMC" = "1" ]; then
  i=0
  for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAVES SQ_BUSY_CYCLES" \
             "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "FETCH_SIZE GRBM_GUI_ACTIVE" \
             "WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    timeout 90 rocprofv3 --kernel-trace --output-format csv --pmc $pmc -d "$OUT/pmc$i" -o long -- $CMD > "$OUT/bench_pmc$i.log" 2>&1
  done
fi
cd "$ROOT"
python tools/prof_long_summary.py "$OUT" > "$OUT/summary.txt"
grep "L=" "$OUT/bench_trace.log"
cat "$OUT/summary.txt"
