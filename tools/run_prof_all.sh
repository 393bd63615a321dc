#!/bin/bash
# every rocprofv3 artefact committed under profiles/ for one round.
# usage (GPU box): bash tools/run_prof_all.sh <tag>
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
# 1. pixel-pass kernels: stats + PMC passes on the C2 workload
bash $ROOT/tools/run_prof.sh $TAG 100000 > $OUT/run_prof.log 2>&1
cd /tmp && export TMPDIR=/tmp
# 2. the bench command itself (kernel stats must agree with bench.py's HIP-event timing)
# (the C4 / C5 legs of a default run are left out here: C5's loglike is the
# same kernel symbol as C2's and would mix into its average)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs > $OUT/bench_under_rocprof.json 2> $OUT/bench_stats.log
# 2b. configs 3, 4 and 5 through the same bench.py, each with its own kernel stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_stats -o run -- python3 $ROOT/bench.py --config C3 --steps 20 --warmup 2 --no-cpu-baseline > $OUT/bench_c3_under_rocprof.json 2> $OUT/c3_stats.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_stats -o run -- python3 $ROOT/bench.py --config C4 --steps 20 --warmup 5 > $OUT/bench_c4_under_rocprof.json 2> $OUT/c4_stats.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -o run -- python3 $ROOT/bench.py --config C5 --steps 50 --warmup 10 > $OUT/bench_c5_under_rocprof.json 2> $OUT/c5_stats.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_c5fetch -o run -- python3 $ROOT/bench.py --config C5 --steps 3 --warmup 1 --settle-steps 0 > /dev/null 2> $OUT/pmc_c5fetch.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_c5write -o run -- python3 $ROOT/bench.py --config C5 --steps 3 --warmup 1 --settle-steps 0 > /dev/null 2> $OUT/pmc_c5write.log
# the instruction / stall counters of the same kernel on the C5 shape (64 x 64 x 16
# gaussians: 64 tiles, 4 tiles per ballot), next to the C2 ones of run_prof.sh
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_c5sq1 -o run -- python3 $ROOT/bench.py --config C5 --steps 3 --warmup 1 --settle-steps 0 > /dev/null 2> $OUT/pmc_c5sq1.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_c5sq2 -o run -- python3 $ROOT/bench.py --config C5 --steps 3 --warmup 1 --settle-steps 0 > /dev/null 2> $OUT/pmc_c5sq2.log
# 3. iterative kernels and batched LM
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/iter_stats -o run -- python3 $ROOT/tools/bench_iter.py 200000 3 > $OUT/iter.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lm_stats -o run -- python3 $ROOT/tools/bench_lm.py 100000 0 > $OUT/lm.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_iter -o run -- python3 $ROOT/tools/bench_iter.py 100000 1 > $OUT/pmc_iter.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_lm -o run -- python3 $ROOT/tools/bench_lm.py 100000 0 > $OUT/pmc_lm.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lmfetch -o run -- python3 $ROOT/tools/bench_lm.py 100000 0 > $OUT/pmc_lmfetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lmwrite -o run -- python3 $ROOT/tools/bench_lm.py 100000 0 > $OUT/pmc_lmwrite.log 2>&1
# 4. round 5: the team form of the lmder step (multi-band and co-elliptical fits)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/team_stats -o run -- python3 $ROOT/tools/lm_advance_share.py 10000 > $OUT/team.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_team -o run -- python3 $ROOT/tools/team_probe.py 20000 9 > $OUT/pmc_team.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_team2 -o run -- python3 $ROOT/tools/team_probe.py 20000 9 > $OUT/pmc_team2.log 2>&1
cd $ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --config C3 --steps 60 > $OUT/bench_c3.json 2>> $OUT/bench.err
python bench.py --config C4 > $OUT/bench_c4.json 2>> $OUT/bench.err
python bench.py --config C5 > $OUT/bench_c5.json 2>> $OUT/bench.err
ls $OUT
