#!/bin/bash
# build, make sure the in-tree library matches the sources, then gpurun:
#   tools/gpu.sh <timeout-seconds> '<command>'
# (a stale or missing library would be rebuilt by every process on the GPU box)
set -e
cd "$(dirname "$0")/.."
make -C ngmix_amd/csrc -j6 2>&1 | grep -v "loop not unrolled\|^ *[0-9]* |\|\^\|warning" | tail -4
python - <<'PY'
import sys
from ngmix_amd import _lib
import os
if not os.path.exists(_lib.LIB_PATH) or _lib.library_is_stale():
    sys.exit("library missing or stale: not going to the GPU")
PY
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
