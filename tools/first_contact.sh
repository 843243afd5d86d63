#!/bin/bash
# One GPU call that catches up on everything rounds 4 and 5 built without hardware, most valuable first (each stage bounded by its
# own timeout; outputs under gpurun_out/fc/).  From the repository root on the GPU box:  bash tools/first_contact.sh [stages...]
# Afterwards: tools/catchup.sh stress inflate rmdup big8 big   (and copy what is to be judged from gpurun_out/ into profiles/)
export TMPDIR=/tmp; O=gpurun_out/fc; mkdir -p $O
# Stage `bisect`: HEAD under SNK_PROVEN_ONLY=1 (generic kernel + LDS histograms: the device sources of the last hardware-green record,
# inside HEAD's own library), then the two trees that HAVE run on an MI355X (tools/ab_trees.sh build, in the build container first) with
# their own tests and bench lines -- run it when `tests` or `quick` are red or slower than 2.611 ms (C2) / 3.403 ms (FULL).
STAGES=${@:-smoke quick tests bench prof unverified}
for st in $STAGES; do
  case $st in
    smoke)  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $O/smoke.txt ;;
    quick)  tools/catchup.sh quick 2>&1 | tee $O/quick.txt ;;
    tests)  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --timeout-method=thread -p no:cacheprovider 2>&1 | tail -25 > $O/tests_verified.txt; tail -3 $O/tests_verified.txt ;;
    bisect) SNK_PROVEN_ONLY=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 | tee $O/bisect_proven_only.txt
            bash tools/ab_trees.sh run 2>&1 | tee $O/bisect_trees.txt ;;
    bench)  timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json ;;
    prof)   tools/catchup.sh prof 2>&1 | tee $O/prof.txt; for t in r4c r4c_c3; do mkdir -p $O/$t; cp gpurun_out/$t/summary.json gpurun_out/$t/*stats*.csv $O/$t/ 2>/dev/null; done ;;
    unverified) timeout 900 python -m pytest tests/test_adapter_fuzz_gpu.py tests/test_rmdup_gpu.py tests/test_gunzip_gpu.py tests/test_long_reads_gpu.py tests/test_cli_gpu.py -m gpu -q --timeout 300 --timeout-method=thread -p no:cacheprovider 2>&1 | tail -40 > $O/tests_unverified.txt; tail -8 $O/tests_unverified.txt ;;
  esac
done
