#!/bin/bash
# On the GPU box: kernel ms of bench workloads under library variants.  tools/variant_bench.sh "<wl> ..." "<name>=<lib>[:ENV=VAL] ..."
ROOT=$(pwd); export TMPDIR=/tmp
for wl in $1; do
  for v in $2; do
    name=${v%%=*}; rest=${v#*=}; lib=${rest%%:*}; envs=""
    if [[ "$rest" == *:* ]]; then envs=${rest#*:}; fi
    ms=$(timeout 150 env ${lib:+SNK_LIB=$ROOT/$lib} ${envs//:/ } python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload $wl | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
    echo "$wl $name $ms"
  done
done
