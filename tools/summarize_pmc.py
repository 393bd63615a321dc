"""summarise rocprofv3 csv output of tools/run_prof.sh:
python tools/summarize_pmc.py gpurun_out/<tag> [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "pixpass_grid_kernel"

stats = os.path.join(root, "stats", "run_kernel_stats.csv")
if os.path.exists(stats):
    print("== kernel stats")
    for row in csv.DictReader(open(stats)):
        if float(row.get("Percentage", 0) or 0) > 0.5:
            print("  %-70s calls %4s avg %10.1f us  %5.1f%%" % (
                row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3,
                float(row["Percentage"])))

for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    f = os.path.join(d, "run_counter_collection.csv")
    if not os.path.exists(f):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat not in k:
            continue
        short = k.split("(")[0].replace("void ngmix::", "")
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", os.path.basename(d))
    for k, cs in acc.items():
        for c, vals in sorted(cs.items()):
            print("  %-34s %-26s n=%d mean %.6g" % (k, c, len(vals), sum(vals) / len(vals)))
