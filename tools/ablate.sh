#!/bin/bash
# Builds ablated variants of the library (profiling only) into soapnuke_amd/abl/ :  tools/ablate.sh 1 2 3 4
set -e
cd "$(dirname "$0")/../soapnuke_amd/csrc"
mkdir -p ../abl
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSNK_ABL=$a -o ../abl/libsnk_abl$a.so \
      snk_filter.cpp snk_generic.hip snk_tiled.hip snk_rmdup.hip -ldl &
done
wait
ls -la ../abl
