#!/bin/bash
# A/B of the SQ counters of one bench config under two builds of the library.
# usage (GPU box): bash tools/pmc_ab.sh <config> <base.so>
CFG=${1:-C5}
BASE=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY"
C2="SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
run() {  # tag
    timeout 300 rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d $OUT/$1_1 -o run -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --settle-steps 0 --no-cpu-baseline > /dev/null 2> $OUT/$1_1.log
    timeout 300 rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d $OUT/$1_2 -o run -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --settle-steps 0 --no-cpu-baseline > /dev/null 2> $OUT/$1_2.log
}
run new
if [ -n "$BASE" ]; then
    export NGMIX_HIP_LIB=$BASE
    run base
    unset NGMIX_HIP_LIB
fi
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("new", "base"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s_*/**/*counter_collection.csv" % (out, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        per = collections.defaultdict(list)
        for (c, disp), v in d.items():
            per[c].append(sum(v))
        tot = sum(sum(v) for v in per.values())
        if "SQ_INSTS_VALU" not in per or max(per["SQ_INSTS_VALU"]) < 1e7:
            continue
        print(tag, k[:70])
        for c in sorted(per):
            v = sorted(per[c])
            print("   %-24s median %.4g  (n=%d)" % (c, v[len(v) // 2], len(v)))
PY
