#!/bin/bash
# kernel-by-kernel timeline of LMBatchFitter.go() on the C3 workload: start /
# end timestamps of every launch of the last calls (rocprofv3 --kernel-trace)
# usage (GPU box): bash tools/lm_trace.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/lm_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 $ROOT/tools/lm_phases.py > $OUT/log.txt 2>&1
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/run_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 60 launches
last = rows[-60:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = t0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +gap %7.1f  dur %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3,
                                                    (e - s) / 1e3, r["Kernel_Name"][:70]))
    prev_end = e
PY
