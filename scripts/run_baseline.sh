cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03a/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03a/bench.json.log 2> gpurun_out/r03a/bench.err
build/lab_gemm 20 > gpurun_out/r03a/lab_gemm.log 2>&1
tail -5 gpurun_out/r03a/pytest.log; cat gpurun_out/r03a/bench.json.log | cut -c1-1500; tail -3 gpurun_out/r03a/lab_gemm.log
