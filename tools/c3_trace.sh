#!/bin/bash
# kernel timeline of the pipelined config-3 bench: gaps between launches
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c3_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 $ROOT/bench.py --config C3 --steps 20 --warmup 2 --settle-steps 40 --no-cpu-baseline > $OUT/log.txt 2>&1
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/run_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-80:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    name = r["Kernel_Name"].replace("void ngmix::", "").replace("ngmix::", "")[:40]
    if True:
        print("%9.1f us  +gap %7.1f  dur %8.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, name))
    busy += e - s
    prev_end = max(prev_end, e)
print("busy %.1f %% of %.1f ms" % (100.0 * busy / (prev_end - t0), (prev_end - t0) / 1e6))
PY
