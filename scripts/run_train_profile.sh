# the fine-tuning step (config 5) timed, then under rocprofv3 --kernel-trace, and one steady-state step by kernel family
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06t; mkdir -p $O
export TMPDIR=/tmp
python scripts/bench_train.py 10 > $O/train_bench.log 2>&1; tail -1 $O/train_bench.log | cut -c1-400
rm -rf $O/rocprof
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/rocprof -- python3 $R/scripts/bench_train.py 6 > $O/train_under_rocprof.log 2>&1)
T=$(ls $O/rocprof/*/*_kernel_trace.csv | head -1)
(tail -1 $O/train_bench.log | cut -c1-420; echo; python scripts/train_step_breakdown.py $T) > $O/r06_train_step_profile.md 2>&1
rm -rf $O/rocprof
head -70 $O/r06_train_step_profile.md | cut -c1-160
