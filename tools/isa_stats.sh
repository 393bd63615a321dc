#!/bin/bash
# per-kernel register counts and instruction mix of the fused pixel-pass kernels
# usage: tools/isa_stats.sh [file.hip] [name-filter]
cd "$(dirname "$0")/../ngmix_amd/csrc"
f=${1:-pixpass.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math \
    --cuda-device-only -S -o /tmp/isa_$$.s "$f" 2>/dev/null
python3 - /tmp/isa_$$.s "${2:-wave_kernel}" <<'PY'
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2]
regs = {}
for m in re.finditer(r'\.name:\s+(\S+)\n((?:\s+\..*\n)+)', s):
    blk = m.group(2)
    sg = re.search(r'\.sgpr_count:\s+(\d+)', blk)
    vg = re.search(r'\.vgpr_count:\s+(\d+)', blk)
    if sg and vg:
        regs[m.group(1)] = (int(sg.group(1)), int(vg.group(1)))
for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
    name = m.group(1)
    if flt not in name:
        continue
    lines = [l.strip() for l in m.group(2).split('\n')]
    lines = [l for l in lines if l and not l.startswith((';', '.', '_Z'))]
    c = lambda p: sum(1 for l in lines if l.startswith(p))
    print(name[:70], 'sgpr/vgpr', regs.get(name), 'v_', c('v_'), 's_', c('s_'),
          'ds_', c('ds_'), 'glob', c('global_'), 'rfl', sum('readfirstlane' in l for l in lines))
PY
rm -f /tmp/isa_$$.s
