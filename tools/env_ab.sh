#!/bin/bash
# config 3 under runtime knobs that could change how device-to-host copies share the GPU with kernels
# usage (GPU box): bash tools/env_ab.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for e in "NGMIX_NOOP=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2" "HSA_ENABLE_SDMA=0" "DEBUG_CLR_LIMIT_BLIT_WG=16" "HIP_FORCE_DEV_KERNARG=0"; do
  env $e python bench.py --config C3 --steps 60 --no-cpu-baseline 2>/dev/null | E="$e" python -c '
import json, os, sys
d = json.loads(sys.stdin.read())
print(os.environ["E"].ljust(30), "%.4g fits/s" % d["value"], "ms/step %.3f" % d["ms_per_step"], "kernel sum %.3f" % d["kernels_ms_sum"])'
done
