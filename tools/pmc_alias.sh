#!/bin/bash
# stall accounting of the fused loglike kernel with and without HBM traffic
# (tools/compute_only.py: real layout, then every stamp aliased onto stamp 0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_alias
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $OUT -o run -- python3 $ROOT/tools/compute_only.py 100000 short > $OUT/log.txt 2>&1
python3 - <<PY
import csv, collections
rows = [r for r in csv.DictReader(open("$OUT/run_counter_collection.csv")) if "pixpass_wave_kernel" in r["Kernel_Name"] and "<0" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
half = len(ids) // 2
for name, sel in (("real", ids[:half]), ("aliased", ids[half:])):
    acc = collections.defaultdict(float)
    for d in sel:
        for k, v in by[d].items():
            acc[k] += v / len(sel) / 1e5
    print(name, {k: round(v, 1) for k, v in sorted(acc.items())})
PY
