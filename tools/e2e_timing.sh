#!/bin/bash
# stage timing of the CLI on the GPU box: tools/e2e_timing.sh <pairs> [host threads ...]
N=${1:-8000000}; shift
ROOT=$(pwd); TMP=$(mktemp -d /dev/shm/snkt_XXXX)
python - "$TMP" "$N" <<'PY'
import sys, os, subprocess
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_e2e
bench_e2e.make_inputs(sys.argv[1], int(sys.argv[2]), ["plain", "gz"])
PY
A="-f AAGTCGGAGGCCAAGCGGTCTTAGGAAGACAA -r AAGTCGGATCGTAGCCATGTCGTTCTGTGAGCCAAGGAGTTG -J -l 10 -q 0.1 -T 16"
for HT in "$@"; do
  for mode in plain gz; do
    if [ $mode = gz ]; then I1=$TMP/r1.fq.gz; I2=$TMP/r2.fq.gz; E=.fq.gz; else I1=$TMP/r1.fq; I2=$TMP/r2.fq; E=.fq; fi
    echo "== host threads $HT, $mode"
    SNK_TIMING=1 SNK_HOST_THREADS=$HT python - $ROOT/soapnuke_amd/SOAPnuke filter -1 $I1 -2 $I2 -C c1$E -D c2$E -o $TMP/out $A <<'PY' 2>&1 | grep -E "timing|Error|wall"
import resource, subprocess, sys, time
t0 = time.time()
subprocess.call(sys.argv[1:])
w = time.time() - t0
u = resource.getrusage(resource.RUSAGE_CHILDREN)
print("wall %.2f s  user %.1f s  sys %.1f s  (%.1f CPUs busy)" % (w, u.ru_utime, u.ru_stime, (u.ru_utime + u.ru_stime) / w))
PY
    rm -rf $TMP/out
  done
done
rm -rf $TMP
