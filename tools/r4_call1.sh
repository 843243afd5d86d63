#!/bin/bash
# One GPU call that catches up on everything round 4 built without hardware, most valuable first (each stage bounded by its own
# timeout; outputs under gpurun_out/r4/).  From the repository root on the GPU box:  bash tools/r4_call1.sh [stages...]
export TMPDIR=/tmp; O=gpurun_out/r4; mkdir -p $O
STAGES=${@:-quick tests bench prof unverified}
for st in $STAGES; do
  case $st in
    quick)  tools/catchup.sh quick 2>&1 | tee $O/quick.txt ;;
    tests)  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --timeout-method=thread -p no:cacheprovider 2>&1 | tail -25 > $O/tests_verified.txt; tail -3 $O/tests_verified.txt ;;
    bench)  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json ;;
    prof)   tools/catchup.sh prof 2>&1 | tee $O/prof.txt; for t in r4c r4c_c3; do mkdir -p $O/$t; cp gpurun_out/$t/summary.json gpurun_out/$t/*stats*.csv $O/$t/ 2>/dev/null; done ;;
    unverified) timeout 900 python -m pytest tests/test_adapter_fuzz_gpu.py tests/test_rmdup_gpu.py tests/test_gunzip_gpu.py tests/test_cli_gpu.py -m gpu -q --timeout 300 --timeout-method=thread -p no:cacheprovider 2>&1 | tail -40 > $O/tests_unverified.txt; tail -8 $O/tests_unverified.txt ;;
  esac
done
