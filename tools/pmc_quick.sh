#!/bin/bash
# quick VALU/SALU/LDS instruction counts + cycles for one library build
# usage: bash tools/pmc_quick.sh <lib.so> <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcq_$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NGMIX_HIP_LIB=$1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT -o run -- python3 $ROOT/tools/prof_pixpass.py 100000 3 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open("$OUT/run_counter_collection.csv")):
    k=row["Kernel_Name"]
    if "pixpass" in k:
        acc[k.split("(")[0][-28:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,cs in acc.items():
    print("$2", k, {c: round(sum(v)/len(v)/4e5,1) for c,v in cs.items()})
PY
