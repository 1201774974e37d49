#!/bin/bash
mkdir -p gpurun_out/r03af
timeout 600 python scripts/exp_flake_attn.py --seconds 25 > gpurun_out/r03af/flake_attn.log 2>&1; grep -E "attn\]|noise\]|Error|error" gpurun_out/r03af/flake_attn.log | head
