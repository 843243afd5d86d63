#!/bin/bash
# tools/catchup.sh <stage> ... : what round 4 could not run because the GPU lease was closed, in the order it matters.  On the GPU box
# (through gpurun), from the repository root.  Stages: tests | stress | quick | bench | prof | inflate | rmdup | big8 | big
# (everything here ran on the CPU emulator of tests/simt before: what is open is speed, and what only the hardware can show)
set -u
ROOT=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
for stage in "$@"; do
  case $stage in
    tests)   # the whole GPU suite, the guarded tests included
      timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/catchup_tests.txt ;;
    stress)
      timeout 300 python tools/stress_parity.py 30 2>&1 | tail -12 | cut -c1-400 | tee gpurun_out/catchup_stress.txt ;;
    bench)
      timeout 1200 python bench.py > gpurun_out/catchup_bench.json 2> gpurun_out/catchup_bench.err; tail -c 6000 gpurun_out/catchup_bench.json ;;
    quick)   # the kernel numbers only
      for wl in c2 c3; do timeout 200 python bench.py --no-cpu-baseline --workload $wl | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', d['roofline']['kernel_ms'], d['roofline']['frac'])"; done ;;
    prof)
      tools/profile.sh r4c c2 > /dev/null 2>&1; tools/profile.sh r4c_c3 c3 > /dev/null 2>&1; python -c "
import json
for t in ('r4c','r4c_c3'):
    d=json.load(open('gpurun_out/%s/summary.json'%t)); print(t, d['trace'], {k:round(v,2) for k,v in d['derived'].items() if 'per_read' in k or 'hbm' in k})" ;;
    inflate)
      for lvl in 1 6; do timeout 300 python tools/bench_gunzip.py 4 $lvl 128 2>&1 | tail -3; done
      timeout 300 python tools/bench_gunzip.py 4 1 32 2>&1 | tail -3
      for v in 0 1; do SNK_DEVICE_INFLATE=$v SNK_TIMING=1 timeout 600 python tools/bench_e2e.py 8000000 16 gz 2>&1 | grep -E "ours|Mreads" | head -3; done ;;
    rmdup)   # one pass against two passes, single end and paired, 4 M reads of 250 positions, 5 % duplicates
      for v in "" 1; do SNK_RMDUP_TWO_PASS=$v timeout 600 python - <<'PY' 2>&1 | tail -2
import os, sys, json, tempfile
sys.path.insert(0, "tools"); import bench_e2e
tmp = tempfile.mkdtemp(prefix="snk_rm_", dir="/dev/shm")
r = bench_e2e.measure(tmp, 4_000_000, 16, ["gz"], c3=False, extra_cfg=["rmdup"], L=250, dup_frac=0.05)
print("two passes" if os.environ.get("SNK_RMDUP_TWO_PASS") else "one pass", json.dumps(r["modes"]["gz"]["ours"]), r["modes"]["gz"].get("report_identical"))
PY
      done ;;
    big)
      timeout 5000 python tools/bench_e2e_big.py 628000000 16 2> gpurun_out/big628.err | tail -1 > gpurun_out/big628.json; tail -5 gpurun_out/big628.err; cat gpurun_out/big628.json ;;
    big8)
      timeout 900 python tools/bench_e2e_big.py 8000000 16 --bounded 2> gpurun_out/big8.err | tail -1 > gpurun_out/big8.json; tail -3 gpurun_out/big8.err; cat gpurun_out/big8.json ;;
  esac
done
