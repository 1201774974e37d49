cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
for hm in "" 1 2; do
  echo "=== LAB_HEADMAJOR=$hm" >> gpurun_out/r03c/lab_attn_hm.log
  env ${hm:+LAB_HEADMAJOR=$hm} LAB_VARIANTS=13 LAB_CASE="L0" build/lab_attn 30 >> gpurun_out/r03c/lab_attn_hm.log 2>&1
done
cat gpurun_out/r03c/lab_attn_hm.log
