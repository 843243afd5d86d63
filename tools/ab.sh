#!/bin/bash
# tools/ab.sh libX.so libY.so ... : kernel time of the bench workload for several builds of the library, interleaved, 3 rounds
# (run on the GPU box; libraries under ab/)
ROOT=$(pwd)
for round in 1 2 3; do
  for l in "$@"; do
    export SNK_LIB=$ROOT/ab/$l
    ms=$(python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
    echo "$l round $round: $ms ms"
  done
done
