#!/bin/bash
# instruction counts / cycles per stamp of the iterative kernels (admom, em)
# usage (GPU box): bash tools/pmc_iter.sh <tag> [nstamps]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
N=${2:-100000}
OUT=$ROOT/gpurun_out/pmci_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT -o run -- python3 $ROOT/tools/bench_iter.py $N 1 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open("$OUT/run_counter_collection.csv")):
    k=row["Kernel_Name"]
    if "admom" in k or "em_" in k:
        acc[k.split("(")[0][-40:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,cs in acc.items():
    print("$1", k, "per stamp:", {c: round(sum(v)/len(v)/$N,1) for c,v in cs.items()})
PY
