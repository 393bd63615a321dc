#!/bin/bash
# round 6: the artefacts committed under profiles/r06_* (GPU box):
#   bash tools/run_prof_r06.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/gputest.log 2>&1
cd /tmp && export TMPDIR=/tmp
# the bench command itself (kernel stats must agree with bench.py's HIP-event timing); the
# other legs are left out: C5's loglike is the same symbol as C2's and would mix into its average
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs > $OUT/bench_under_rocprof.json 2> $OUT/bench_stats.log
# C3 as the bench runs it: batches PIPELINED on the host (go_stream): kernels of neighbouring
# batches overlap, the sum of average durations per fit exceeds ms_per_step
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_stats -o run -- python3 $ROOT/bench.py --config C3 --steps 20 --warmup 2 --no-cpu-baseline > $OUT/bench_c3_under_rocprof.json 2> $OUT/c3_stats.log
# C3 UN-pipelined: one synchronous LMBatchFitter.go() at a time (tools/bench_lm.py)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lm_stats -o run -- python3 $ROOT/tools/bench_lm.py 100000 0 > $OUT/lm.log 2>&1
# co-elliptical / multi-band fits: the team step and the precise covariance pass
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/team_stats -o run -- python3 $ROOT/tools/lm_advance_share.py 10000 > $OUT/team.log 2>&1
cd $ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --config C3 --steps 60 > $OUT/bench_c3.json 2>> $OUT/bench.err
python tools/bench_boot_psf.py > $OUT/boot_psf.log 2>&1
ls $OUT
tail -3 $OUT/gputest.log
