#!/bin/bash
# debugging helper: run one pytest node, keep its tmp dir, diff report files
python -m pytest "$@" -x -q --basetemp=/tmp/dbgtmp 2>&1 | tail -5
for d in /tmp/dbgtmp/*/; do
  for o in ours host; do
    [ -d $d/$o ] || continue
    for f in $d/ref/*.txt; do
      b=$(basename $f)
      if ! cmp -s $f $d/$o/$b; then echo "=== $o $b"; diff $f $d/$o/$b | head -20; fi
    done
  done
done
