#!/bin/bash
# rocprof of the fine-tuning step (config 5), per-kernel-family totals
mkdir -p gpurun_out/r03o
O=$PWD/gpurun_out/r03o
python scripts/bench_train.py 6 > $O/bench_train.json 2> $O/bench_train.err
cat $O/bench_train.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o train -- python3 $R/scripts/bench_train.py 6 > $O/rocprof.log 2>&1
cd $R
CSV=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python scripts/rocprof_summary.py $CSV 9 > $O/train_summary.md
cp $CSV $O/train_kernel_stats.csv
rm -rf $O/prof
head -50 $O/train_summary.md
