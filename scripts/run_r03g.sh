cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g; : > gpurun_out/r03g/ab.log
for rep in 1 2; do
echo "== new reduce (rep $rep)" >> gpurun_out/r03g/ab.log
LAB_COLSUM=1 LAB_ONLY="conv" build/lab_gemm 20 2>&1 | grep "split-K\|TOTAL" >> gpurun_out/r03g/ab.log
echo "== old reduce (rep $rep)" >> gpurun_out/r03g/ab.log
LD_LIBRARY_PATH=$PWD/build/variants/old LAB_COLSUM=1 LAB_ONLY="conv" build/lab_gemm 20 2>&1 | grep "split-K\|TOTAL" >> gpurun_out/r03g/ab.log
done
cat gpurun_out/r03g/ab.log
