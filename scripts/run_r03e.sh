cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_pipeline.py tests/test_gpu_kernels.py -x -q -k "sampler or ddim or cfg or pipeline or prompt or captured or graph" 2>&1 | tail -8 > gpurun_out/r03e/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train > gpurun_out/r03e/bench.json.log 2> gpurun_out/r03e/bench.err
cat gpurun_out/r03e/pytest.log; cut -c1-400 gpurun_out/r03e/bench.json.log; tail -3 gpurun_out/r03e/bench.err
