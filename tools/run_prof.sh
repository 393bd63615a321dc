#!/bin/bash
# rocprofv3 passes for the pixel-pass kernels on the C2 workload.
# usage (on the GPU box): bash tools/run_prof.sh <tag> [nstamps]
# Output: gpurun_out/<tag>/{stats,pmc_*}/...csv   (copy summaries to profiles/)
TAG=${1:-prof}
N=${2:-100000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/tools/prof_pixpass.py $N 5 > $OUT/stats.log 2>&1
pass() {
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o run -- python3 $ROOT/tools/prof_pixpass.py $N 3 > $OUT/pmc_$name.log 2>&1
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
pass sq3 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_IFETCH SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
find $OUT -name "*.csv" | head -40
