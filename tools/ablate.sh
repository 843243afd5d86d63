#!/bin/bash
# Builds ablated variants of the library (profiling only) into ab/ (git-ignored *.so; travels to the GPU box) :  tools/ablate.sh 1 2 3 4
# (the 129..160-position instances of the tiled kernel only: SNK_ONLY_NW=5)
set -e
cd "$(dirname "$0")/../soapnuke_amd/csrc"
mkdir -p ../../ab
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSNK_ABL=$a -DSNK_ONLY_NW=5 -o ../../ab/libsnk_abl$a.so \
      snk_filter.cpp snk_generic.hip snk_tiled.hip snk_rmdup.hip snk_contam.hip snk_long.hip snk_fastq.hip snk_gzip.hip snk_inflate.hip -ldl &
done
wait
ls -la ../../ab
