#!/bin/bash
# one-off counter passes over one bench config: which unit a kernel waits for.
# usage (GPU box): bash tools/pmc_probe.sh <config> "<counters pass 1>" ["<counters pass 2>" ...]
CFG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o run -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --settle-steps 0 --no-cpu-baseline > /dev/null 2> $OUT/p$i.log || tail -5 $OUT/p$i.log
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob("%s/p*/**/*counter_collection.csv" % out, recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
for k, d in acc.items():
    if "ngmix" not in k:
        continue
    print(k[:90])
    for c in sorted(d):
        v = sorted(d[c].values())
        print("   %-28s median %.5g  (n=%d)" % (c, v[len(v) // 2], len(v)))
PY
