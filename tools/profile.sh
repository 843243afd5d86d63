#!/bin/bash
# Profiles a bench workload on the GPU box (run through gpurun from the repo root):
#   tools/profile.sh <tag> [workload]        workload: c2 (headline, default) | c3 | c2var | c3var  (bench.py --workload)
# pass 0: rocprofv3 --kernel-trace --stats of `bench.py` (no cpu baseline)
# pass 1..4: --pmc counter passes (own runs, kernel-trace only), summarised by tools/pmc_summary.py
# Results land under gpurun_out/<tag>/ ; copy the summaries to profiles/ and commit them.
set -u
TAG=${1:-prof}
WL=${2:-c2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 20 --warmup 4 --no-cpu-baseline --workload $WL"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o tiled -- $CMD > "$OUT/bench_trace.log" 2>&1
i=0
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $pmc -d "$OUT/pmc$i" -o tiled -- $CMD --steps 6 > "$OUT/bench_pmc$i.log" 2>&1
done
cd "$ROOT"
python tools/pmc_summary.py "$OUT" > "$OUT/summary.json"
cp $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv" 2>/dev/null
cat "$OUT/summary.json"
