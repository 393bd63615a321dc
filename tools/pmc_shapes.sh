#!/bin/bash
# usage (on the GPU box): bash tools/pmc_shapes.sh <tag> [lib.so]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcs_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export NGMIX_HIP_LIB=$2
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT -o run -- python3 $ROOT/tools/pmc_shapes.py 20000 2 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, collections
rows = [r for r in csv.DictReader(open("$OUT/run_counter_collection.csv"))
        if "pixpass_wave_kernel<0" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
shapes = ["8x8", "8x16", "16x16", "16x32", "32x32", "48x48"]
ids = sorted(by)
for i, d in enumerate(ids):
    c = by[d]
    print("$1", shapes[i // 2] if i // 2 < len(shapes) else "?", "per-wave:",
          {k: round(v / 20000, 1) for k, v in sorted(c.items())})
PY
