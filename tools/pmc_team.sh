#!/bin/bash
# instruction counts / busy cycles of lm_advance_team_kernel per fit
# usage (on the GPU box): bash tools/pmc_team.sh <tag> [nobj] [nband]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcteam_$1
NOBJ=${2:-20000}
NBAND=${3:-9}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT -o run -- python3 $ROOT/tools/team_probe.py $NOBJ $NBAND > $OUT/log.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -o run2 -- python3 $ROOT/tools/team_probe.py $NOBJ $NBAND >> $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in rows:
        k = row["Kernel_Name"]
        if "lm_advance" in k:
            acc[k.split("(")[0][-40:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        # the largest value of each counter = a launch with every fit active
        print("$1", k, {c: round(max(v) / $NOBJ, 1) for c, v in cs.items()}, "(per fit, fullest launch)")
PY
