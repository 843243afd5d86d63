# phase split of the C3 (full trim + filter) configuration: tools/ablate.sh 1 2 3 4 5 first
ROOT=$(pwd)
for a in ${ABLS:-0 1 2 3 4 5}; do
  if [ $a = 0 ]; then unset SNK_LIB; else export SNK_LIB=$ROOT/ab/libsnk_abl$a.so; fi
  echo "abl $a: $(python tools/bench_configs.py 'C3 full' | tail -1)"
done
