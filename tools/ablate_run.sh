#!/bin/bash
# On the GPU box: time + SQ counters of each ablated build (tools/ablate.sh) and of the shipped library.
#   tools/ablate_run.sh <workload> <abl> ...      (workload: bench.py --workload)
ROOT=$(pwd); WL=$1; shift; OUT=$ROOT/gpurun_out/abl_$WL; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for a in 0 "$@"; do
  if [ $a = 0 ]; then unset SNK_LIB; else export SNK_LIB=$ROOT/ab/libsnk_abl$a.so; fi
  ms=$(python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload $WL | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY \
     -d $OUT/p$a -o t -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $WL > $OUT/log$a 2>&1
  python - $OUT/p$a $a $ms <<'PY' | tee -a $OUT/table.txt
import csv,glob,sys,os
acc={}
for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if "snk_tiled_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"],{}).setdefault(r["Dispatch_Id"],0.0)
            acc[r["Counter_Name"]][r["Dispatch_Id"]]+=float(r["Counter_Value"])
m={k:sum(v.values())/len(v) for k,v in acc.items()}
wc=m.get("SQ_WAVE_CYCLES",1)
print("abl",sys.argv[2],"ms",sys.argv[3],"valu/read %.1f salu/read %.1f lds/read %.1f | of wave cycles: valu %.3f any %.3f wait_any %.3f wait_inst %.3f"%(
 m.get("SQ_INSTS_VALU",0)/20e6,m.get("SQ_INSTS_SALU",0)/20e6,m.get("SQ_INSTS_LDS",0)/20e6,m.get("SQ_ACTIVE_INST_VALU",0)/wc,m.get("SQ_ACTIVE_INST_ANY",0)/wc,m.get("SQ_WAIT_ANY",0)/wc,m.get("SQ_WAIT_INST_ANY",0)/wc))
PY
  rm -rf $OUT/p$a
done
