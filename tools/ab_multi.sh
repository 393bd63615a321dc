#!/bin/bash
# interleaved timing of several builds of libngmix_hip.so on the C2 workload,
# then VALU/SALU/LDS instruction counts of the last one
# usage: bash tools/ab_multi.sh <rounds> <libA.so> <libB.so> ...
R=$1; shift
for i in $(seq 1 $R); do
  for L in "$@"; do
    NGMIX_HIP_LIB=$PWD/$L timeout 180 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', {k: round(v,4) for k,v in d['kernels_ms'].items()}, round(d['loglike_stamp_evals_per_s_per_gpu']/1e6,1), 'bad', d['bad_status'])"
  done
done


