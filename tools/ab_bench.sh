#!/bin/bash
# tools/ab_bench.sh "<workloads>" "<lib names under ab/ | ->" [rounds] : kernel ms of bench workloads under library variants, interleaved
# ("-" = the shipped library).  Run on the GPU box.
ROOT=$(pwd); export TMPDIR=/tmp
for round in $(seq 1 ${3:-2}); do
  for wl in $1; do
    for v in $2; do
      if [ "$v" = "-" ]; then unset SNK_LIB; else export SNK_LIB=$ROOT/ab/libsnk_$v.so; fi
      r=$(timeout 150 python $ROOT/bench.py --steps 20 --warmup 4 --no-cpu-baseline --workload $wl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['roofline']['frac'])")
      echo "$wl $v $r"
    done
  done
done
