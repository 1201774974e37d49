#!/bin/bash
mkdir -p gpurun_out/r03s
O=$PWD/gpurun_out/r03s
R=$PWD
timeout 1200 python -m pytest tests/test_train_kernels.py tests/test_gpu_train.py tests/test_fstext.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python scripts/bench_train.py 10 > $O/bench_train.json 2> $O/bench_train.err; cut -c1-330 $O/bench_train.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o train -- python3 $R/scripts/bench_train.py 6 > $O/rocprof.log 2>&1
cd $R
CSV=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python scripts/rocprof_summary.py $CSV 9 > $O/train_summary.md
cp $CSV $O/train_kernel_stats.csv
rm -rf $O/prof
grep -v "at::native\|rocclr\|elementwise_kernel_with_index" $O/train_summary.md | head -40
