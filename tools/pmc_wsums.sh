#!/bin/bash
# VALU instruction count / busy cycles of the fused weighted-sums kernels
# usage (on the GPU box): bash tools/pmc_wsums.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcws_$1
mkdir -p $OUT
cat > /tmp/ws_driver.py <<PY
import sys, os
sys.path.insert(0, "$ROOT")
import numpy as np, torch, bench
from ngmix_amd.batch import GMixBatch
n = 100000
sb, gm, pars = bench.make_workload(n, seed=1000, device=torch.device("cuda", 0))
wt, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.6, 1.0], (n, 1)), "gauss")
wt.set_norms()
for _ in range(3):
    res, st = sb.weighted_sums(wt, maxrad=1.0e9)
torch.cuda.synchronize()
print("done")
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT -o run -- python3 /tmp/ws_driver.py > $OUT/log.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -o run -- python3 /tmp/ws_driver.py > $OUT/log2.txt 2>&1
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open("$OUT/run_counter_collection.csv")):
    k = row["Kernel_Name"]
    if "wsums" in k:
        acc[k.split("(")[0][-30:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print("$1", k, {c: round(max(v) / 1e5, 1) for c, v in cs.items()}, "(per stamp)")
for row in csv.DictReader(open(glob.glob("$OUT/st/*kernel_stats.csv")[0])):
    if "wsums" in row["Name"]: print(row["Name"][:50], row["Calls"], row["AverageNs"])
PY
tail -3 $OUT/log.txt
