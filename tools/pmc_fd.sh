#!/bin/bash
# instruction counts / busy cycles of the forward-difference LM kernel ('bdf' fits)
# usage (on the GPU box): bash tools/pmc_fd.sh <tag> [model]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcfd_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT -o run -- python3 $ROOT/tools/bench_lm_fd.py 20000 ${2:-bdf} > $OUT/log.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -o run -- python3 $ROOT/tools/bench_lm_fd.py 20000 ${2:-bdf} > $OUT/logb.txt 2>&1
python3 - <<PY
import csv, collections, glob
for f in ("$OUT/run_counter_collection.csv", "$OUT/b/run_counter_collection.csv"):
    try:
        rows = list(csv.DictReader(open(f)))
    except Exception as e:
        print("missing", f); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in rows:
        k = row["Kernel_Name"]
        if "lm_eval_fd" in k:
            acc[k.split("(")[0][-30:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print("$1", k, {c: round(max(v) / 2e4, 1) for c, v in cs.items()}, "(per stamp, jacobian launch)")
PY
