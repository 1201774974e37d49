cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O; : > $O/out.log
python -m pytest tests/test_gpu_kernels.py -x -q -k "splitk or colsum or split_k or conv3x3" 2>&1 | tail -4 >> $O/out.log
for rep in 1 2; do
echo "== two launches (rep $rep)" >> $O/out.log
LAB_COLSUM=1 build/lab_gemm 20 2>&1 | grep "split-K\|TOTAL" >> $O/out.log
echo "== in-launch reduction (rep $rep)" >> $O/out.log
LAB_INLAUNCH=1 LAB_COLSUM=1 build/lab_gemm 20 2>&1 | grep "split-K\|TOTAL" >> $O/out.log
done
python scripts/exp_flake2.py --cotenant model --launches 1000 --seconds 15 --cases ff2mid:384:1280:5120:br:0,l3short:384:1280:2560:b:0 2>&1 | grep -v amdgpu >> $O/out.log
python scripts/exp_flake.py --colsums 1 --cotenant 1 --trace 0 --iters 2000 2>&1 | grep "RESULT" >> $O/out.log
python scripts/exp_flake.py --colsums 1 --cotenant 1 --trace 0 --iters 300 --batch 2 --frames 6 --latent 32 2>&1 | grep "RESULT" >> $O/out.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | cut -c1-330 >> $O/out.log
cat $O/out.log
