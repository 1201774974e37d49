cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03f
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03f/trace -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-train > $R/gpurun_out/r03f/bench_traced.log 2>&1
cd $R
F=$(ls gpurun_out/r03f/trace/*/*_kernel_trace.csv | head -1)
python scripts/trace_gaps.py $F > gpurun_out/r03f/step_timeline.md 2>&1
cat gpurun_out/r03f/step_timeline.md
rm -rf gpurun_out/r03f/trace
