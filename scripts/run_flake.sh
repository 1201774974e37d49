cd $GRAFT_REPO_ROOT
L=gpurun_out/flake9.log; : > $L
F="RESULT\|first differing op ("
python scripts/exp_flake.py --colsums 1 --cotenant 1 --trace 1 --iters 3000 2>&1 | grep "$F" >> $L
python scripts/exp_flake.py --colsums 1 --cotenant 1 --trace 0 --iters 3000 2>&1 | grep "$F" >> $L
python scripts/exp_flake.py --colsums 1 --cotenant 1 --trace 0 --iters 1500 --batch 2 --frames 3 --latent 32 2>&1 | grep "$F" >> $L
cat $L
python -m pytest tests/test_gpu_cotenant.py tests/test_gpu_kernels.py -x -q -k "cotenant or colsum or rotary" 2>&1 | tail -15
