#!/bin/bash
# step timeline of the current build: rocprofv3 --kernel-trace of the bench command, cut into steps -> gpurun_out/tl/
R=$(pwd); O=$R/gpurun_out/tl; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train > $O/bench_under_rocprof.json.log 2>&1)
T=$(ls $O/rocprof/*/*_kernel_trace.csv | head -1)
python scripts/trace_gaps.py $T > $O/step_timeline.md 2>&1
rm -rf $O/rocprof
