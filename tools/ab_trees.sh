#!/bin/bash
# tools/ab_trees.sh [build|run] : A/B trees of the two builds that HAVE run on an MI355X, so that a red or slow first
# contact of HEAD can be told apart from a red environment inside one gpurun call instead of by `git checkout` + rebuild on the box:
#   abl/hw_d12b8ba  the last build that executed on hardware (round 4, builder-run profiles r04_c2_*: 2.611 ms / 0.303 of peak)
#   abl/r03_5671319 round 3's HEAD, the last green driver GPUTEST (623 passed) and the last driver BENCH line (2.691 ms per step)
# `build` (in the build container): git archive of each commit into abl/<tag>/, built there with its OWN __graft_entry__.build()
#   (library, CLI, oracle, oracle/_ref) -- whole trees, because each commit's tests and bench.py belong to its own C ABI.
#   abl/ is git-ignored (built artefacts + old sources already in the history) and NOT gpurun-ignored: it travels to the GPU box.
# `run` (on the GPU box, from the repository root): each tree's own tests/test_gpu_parity.py and kernel-only bench line, then HEAD's,
#   into gpurun_out/ab/<tag>.txt -- green old tree + red HEAD = HEAD's kernels; red old tree = the box.
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd)
TREES="hw_d12b8ba:d12b8ba r03_5671319:5671319"
case ${1:-build} in
  build)
    for t in $TREES; do
      tag=${t%%:*}; sha=${t##*:}
      if [ -f abl/$tag/soapnuke_amd/libsnk_filter.so ]; then echo "$tag: built"; continue; fi
      rm -rf abl/$tag; mkdir -p abl/$tag
      git archive $sha | tar -x -C abl/$tag
      rm -rf abl/$tag/profiles abl/$tag/gpurun_out abl/$tag/*.md abl/$tag/*_r0*.json
      (cd abl/$tag && python -c "import __graft_entry__ as g; g.build()" > build.log 2>&1) || { echo "$tag: build failed (abl/$tag/build.log)"; continue; }
      rm -rf abl/$tag/soapnuke_amd/csrc/build abl/$tag/oracle/_ref/obj abl/$tag/oracle/_ref/pic
      echo "$tag: $(ls -la abl/$tag/soapnuke_amd/libsnk_filter.so | awk '{print $5}') bytes"
    done ;;
  run)
    export TMPDIR=/tmp; mkdir -p gpurun_out/ab
    for d in abl/hw_d12b8ba abl/r03_5671319 .; do
      tag=$(basename $d); [ $d = . ] && tag=HEAD
      [ -f $d/soapnuke_amd/libsnk_filter.so ] || { echo "$tag: not built"; continue; }
      ( cd $d
        timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout 120 2>&1 | tail -3
        for wl in c2 c3; do
          timeout 300 python bench.py --no-cpu-baseline --workload $wl 2>/dev/null | tail -1 | python -c "
import sys, json
t = sys.stdin.read().strip()
if not t.startswith('{'):
    print('$tag', '$wl', 'no bench line (the run died: an older bench.py has no child for its timed region)'); sys.exit(0)
d = json.loads(t); r = d.get('roofline') or {}
print('$tag', '$wl', 'kernel_ms', r.get('kernel_ms'), 'frac', r.get('frac'), 'ms_per_step', d.get('ms_per_step'), d.get('error', ''))"
        done ) > gpurun_out/ab/$tag.txt 2>&1
      echo "== $tag"; cat gpurun_out/ab/$tag.txt
    done
    # ... and HEAD's own bench.py on each OLD library (SNK_LIB: soapnuke_amd/abi.py stubs the entry points an older library lacks): the same
    # workload code and timing around the old kernels -- if the parameter block's struct_size is still accepted there
    for d in abl/hw_d12b8ba abl/r03_5671319; do
      [ -f $d/soapnuke_amd/libsnk_filter.so ] || continue
      SNK_LIB=$ROOT/$d/soapnuke_amd/libsnk_filter.so timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
t = sys.stdin.read().strip()
d = json.loads(t) if t.startswith('{') else {}
r = d.get('roofline') or {}
print('HEAD bench.py on the library of $(basename $d):', 'kernel_ms', r.get('kernel_ms'), 'frac', r.get('frac'), (d.get('error') or '')[:120])" | tee -a gpurun_out/ab/HEAD.txt
    done ;;
esac
