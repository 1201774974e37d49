set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06t; mkdir -p $O
export TMPDIR=/tmp
python scripts/bench_train.py 10 > $O/train_bench.log 2>&1; tail -1 $O/train_bench.log | cut -c1-400
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 $R/scripts/bench_train.py 6 > $O/train_under_rocprof.log 2>&1)
S=$(ls $O/rocprof/*/*_kernel_stats.csv | head -1)
python scripts/rocprof_summary.py $S 9 > $O/r06_train_step_profile.md 2>&1
head -60 $O/r06_train_step_profile.md | cut -c1-160
