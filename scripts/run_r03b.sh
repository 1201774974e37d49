cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_kernels.py -x -q -k "return_attn or vae_decode_full or alternate_frame or attention or full_size_step_matches" 2>&1 | tail -8 > gpurun_out/r03b/pytest.log
build/lab_attn 30 > gpurun_out/r03b/lab_attn.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train > gpurun_out/r03b/bench.json.log 2> gpurun_out/r03b/bench.err
cat gpurun_out/r03b/pytest.log; grep -A3 "spatial L0\|temporal L0" gpurun_out/r03b/lab_attn.log | head -20; cut -c1-1800 gpurun_out/r03b/bench.json.log
