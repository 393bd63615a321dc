#!/bin/bash
# CPU sanitizer run (race detection / sanitizers), never on
# the GPU box: AddressSanitizer + UndefinedBehaviorSanitizer builds of
#   * the oracle                        (make -C oracle asan)
#   * the HOST side of libngmix_hip.so  (make -C ngmix_amd/csrc asan: --cuda-host-only)
# and the CPU test files that drive them through ctypes -- the C-ABI host entry
# points (seam forms' host arithmetic, record layouts, error paths), the lmder
# iteration in both forms (ngmix_lm_advance_host) and the oracle against the
# reference's golden vectors.
# usage: bash tools/run_sanitizers.sh [log file]     (default profiles/r06_sanitizers.log)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$ROOT/profiles/r06_sanitizers.log}
CLANG=/opt/rocm/lib/llvm/bin/clang
make -C "$ROOT/oracle" asan > /dev/null
make -C "$ROOT/ngmix_amd/csrc" -j4 asan > /dev/null 2>&1
RT=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
{
  echo "# $(date -u +%FT%TZ)  $(git rev-parse --short HEAD)  tools/run_sanitizers.sh"
  echo "# LD_PRELOAD=$RT  NGMIX_HIP_LIB=ngmix_amd/libngmix_hip_asan.so  NGMIX_ORACLE_LIB=oracle/libngmix_oracle_asan.so"
  # (CPython itself is not instrumented and leaks by design: no leak check;
  # the kernel-resource / ISA-hazard tests read DEVICE code objects, which a
  # host-only library does not have)
  LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  NGMIX_HIP_LIB=$ROOT/ngmix_amd/libngmix_hip_asan.so \
  NGMIX_ORACLE_LIB=$ROOT/oracle/libngmix_oracle_asan.so \
  python -m pytest tests/test_cabi_host.py tests/test_lm_core.py tests/test_oracle_golden.py \
      tests/test_host_logic.py -q -m "not gpu" -p no:cacheprovider \
      -k "not resource_guard and not hazard_guard and not every_declared_symbol and not library_matches_sources" 2>&1 | tail -25
} | tee "$LOG"
