#!/bin/bash
# tools/pmc_quick.sh <tag> <command...> : SQ instruction mix of the tiled kernel under an arbitrary command
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
   -d $OUT -o q -- "$@" > $OUT/log 2>&1
python - $OUT <<'PY'
import csv,glob,sys,os
acc={}; dur={}
for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if "snk_tiled_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"],{}).setdefault(r["Dispatch_Id"],0.0)
            acc[r["Counter_Name"]][r["Dispatch_Id"]]+=float(r["Counter_Value"])
m={k:sum(v.values())/len(v) for k,v in acc.items()}
wc=m.get("SQ_WAVE_CYCLES",1)
print({k:round(v/1e6,1) for k,v in m.items()}, "Minstr per launch; valu_frac %.3f wait_any %.3f wait_inst %.3f"%(m.get("SQ_ACTIVE_INST_VALU",0)/wc, m.get("SQ_WAIT_ANY",0)/wc, m.get("SQ_WAIT_INST_ANY",0)/wc))
PY
