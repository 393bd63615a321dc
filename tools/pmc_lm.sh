#!/bin/bash
# VALU / SALU / LDS instruction counts and busy cycles of the batched-LM kernels
# on the C3 workload (first launch of each fit = all stamps active)
# usage (on the GPU box): bash tools/pmc_lm.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmclm_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT -o run -- python3 $ROOT/bench.py --config C3 --steps 2 --warmup 1 --settle-steps 0 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/run_counter_collection.csv")))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in rows:
    k = row["Kernel_Name"]
    if "lm_" in k:
        acc[k.split("(")[0][-30:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    # the largest value of each counter = a launch with every fit active
    print("$1", k, {c: round(max(v) / 1e5, 1) for c, v in cs.items()}, "(per object, full launch)")
PY
