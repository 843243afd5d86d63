#!/bin/bash
# tools/variant_build.sh <name> [-DFLAG ...] : an A/B build of the library into ab/libsnk_<name>.so (git-ignored, travels to the GPU
# box); the 129..160-position instances of the tiled kernel only (SNK_ONLY_NW=5) unless -DSNK_ALL_NW is among the flags
set -e
name=$1; shift
cd "$(dirname "$0")/../soapnuke_amd/csrc"
mkdir -p ../../ab
only="-DSNK_ONLY_NW=5"
for f in "$@"; do if [ "$f" = "-DSNK_ALL_NW" ]; then only=""; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $only "$@" -o ../../ab/libsnk_$name.so \
    snk_filter.cpp snk_generic.hip snk_tiled.hip snk_rmdup.hip snk_contam.hip snk_long.hip snk_fastq.hip snk_gzip.hip snk_inflate.hip -ldl 2>&1 | grep -E "error" || true
ls -la ../../ab/libsnk_$name.so
