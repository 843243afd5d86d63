#!/bin/bash
# A/B of the CLI's host pipeline on the GPU box: tools/e2e_ab.sh <pairs> "<name>:<ENV=VAL,...>" ...   (name "dflt:" = no extra environment)
# plain -> plain and .gz -> .gz, BASELINE configs[1] parameters, stage clocks (SNK_TIMING) + wall / user / sys per run
N=${1:-8000000}; shift
ROOT=$(pwd); TMP=$(mktemp -d /dev/shm/snkt_XXXX)
python - "$TMP" "$N" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_e2e
bench_e2e.make_inputs(sys.argv[1], int(sys.argv[2]), ["plain", "gz"])
PY
A="-f AAGTCGGAGGCCAAGCGGTCTTAGGAAGACAA -r AAGTCGGATCGTAGCCATGTCGTTCTGTGAGCCAAGGAGTTG -J -l 10 -q 0.1 -T 16"
for V in "$@"; do
  name=${V%%:*}; envs=${V#*:}
  for mode in plain gz; do
    if [ $mode = gz ]; then I1=$TMP/r1.fq.gz; I2=$TMP/r2.fq.gz; E=.fq.gz; else I1=$TMP/r1.fq; I2=$TMP/r2.fq; E=.fq; fi
    for rep in 1 2; do
    echo "== $name $mode run $rep"
    env SNK_TIMING=1 ${envs//,/ } python - $ROOT/soapnuke_amd/SOAPnuke filter -1 $I1 -2 $I2 -C c1$E -D c2$E -o $TMP/out_$name $A <<'PY' 2>&1 | grep -E "timing|Error|wall"
import resource, subprocess, sys, time
t0 = time.time()
subprocess.call(sys.argv[1:])
w = time.time() - t0
u = resource.getrusage(resource.RUSAGE_CHILDREN)
print("wall %.2f s  user %.1f s  sys %.1f s  (%.1f CPUs busy)" % (w, u.ru_utime, u.ru_stime, (u.ru_utime + u.ru_stime) / w))
PY
    done
    (cd $TMP/out_$name && if [ $mode = gz ]; then zcat c1$E | md5sum; zcat c2$E | md5sum; else md5sum < c1$E; md5sum < c2$E; fi; md5sum *.txt | md5sum)
    rm -rf $TMP/out_$name
  done
done
rm -rf $TMP
