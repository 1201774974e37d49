cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
LAB_ONLY=qkv build/lab_gemm 30 > gpurun_out/r03d/lab_qkv.log 2>&1
python -m pytest tests/test_gpu_kernels.py -x -q -k "rotary or head_major or attention_d40" 2>&1 | tail -3 >> gpurun_out/r03d/lab_qkv.log
cat gpurun_out/r03d/lab_qkv.log
