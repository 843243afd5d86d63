# times the library variants named on the command line (ab/libsnk_<name>.so) with bench.py
ROOT=$(pwd)
for a in "$@"; do
  export SNK_LIB=$ROOT/ab/libsnk_$a.so
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline | tail -1 | python -c "import sys,json; print('variant', '$a', json.loads(sys.stdin.read())['roofline']['kernel_ms'])"
done
