#!/bin/bash
# round 4: fp16 VAE path -- kernel tests, VAE parity (decode + encode), pipeline / lifecycle users of the VAE
mkdir -p gpurun_out/r04
timeout 1800 python -m pytest tests/test_gpu_f16.py tests/test_gpu_unet.py -k "f16 or vae or decode" tests/test_vae_encode.py tests/test_gpu_lifecycle.py tests/test_pipeline.py -x -q -s 2>&1 | grep -E "parity|passed|failed|Error|error|assert" | tail -40 > gpurun_out/r04/tests_f16.log
cat gpurun_out/r04/tests_f16.log | tail -30
