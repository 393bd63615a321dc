#!/bin/bash
# interleaved A/B timing of two builds of libngmix_hip.so on the C2 workload
# usage: bash tools/ab_bench.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq 1 $R); do
  for L in $A $B; do
    NGMIX_HIP_LIB=$L python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['kernels_ms'], round(d['loglike_stamp_evals_per_s_per_gpu']/1e6,1))"
  done
done
