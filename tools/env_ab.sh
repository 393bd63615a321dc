#!/bin/bash
# one bench config under a list of environment settings, alternating, twice
# usage (GPU box): bash tools/env_ab.sh <config> "A=1" "B=1" ...   ("NONE=1" = the default build)
CFG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
for e in "$@"; do
  env $e python bench.py --config $CFG --no-cpu-baseline 2>/dev/null | tail -1 | E="$e" python -c '
import json, os, sys
d = json.loads(sys.stdin.read())
r = d["roofline"]
print(os.environ["E"].ljust(28), "%.4g" % d["value"], "ms/step %.4f" % d["ms_per_step"], r.get("kernel", "")[:40], "frac %.4f" % r["frac"], "launch %.4f ms" % r.get("avg_launch_ms", 0))'
done
done
