# every measurement that DESIGN.md / profiles/ quote for the round-6 build, in one gpurun call
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06final
mkdir -p $O
cd $R
# 1. PMC passes over one eager step (separate passes, no trace domains beside them)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/scripts/pmc_step.py > $O/pmc_$c.log 2>&1)
done
(cd /tmp && rocprofv3 --pmc MfmaUtil VALUBusy LdsUtil OccupancyPercent --output-format csv -d $O/pmc_util -- python3 $R/scripts/pmc_step.py > $O/pmc_util.log 2>&1)
python scripts/pmc_summary.py $(ls $O/pmc_FETCH_SIZE/*/*_counter_collection.csv | head -1) $(ls $O/pmc_WRITE_SIZE/*/*_counter_collection.csv | head -1) > $O/r06_pmc_traffic.json 2> $O/pmc_summary.err
python scripts/pmc_util_summary.py $(ls $O/pmc_util/*/*_counter_collection.csv | head -1) > $O/r06_pmc_utilisation.json 2>> $O/pmc_summary.err
cp $O/r06_pmc_traffic.json profiles/r06_pmc_traffic.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_util
# 2. the bench line (default command), with the traffic file of THIS build in place
python bench.py > $O/r06_bench_final.json.log 2> $O/bench.err
# 3. the same command under rocprofv3 --kernel-trace --stats
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train > $O/r06_bench_under_rocprof.json.log 2>&1)
S=$(ls $O/rocprof/*/*_kernel_stats.csv | head -1)
cp $S $O/r06_rocprof_kernel_stats.csv
T=$(ls $O/rocprof/*/*_kernel_trace.csv | head -1)
python scripts/trace_gaps.py $T > $O/r06_step_timeline.md 2>&1
python scripts/rocprof_summary.py $S 133 > $O/r06_rocprof_summary.md 2>&1
rm -rf $O/rocprof
# 4. per-shape tables
build/lab_gemm 20 > $O/r06_lab_gemm_step_table.log 2>&1
LAB_MMUL=4 build/lab_gemm 10 > $O/r06_lab_gemm_step_table_config4.log 2>&1
build/lab_attn 30 > $O/r06_lab_attn.log 2>&1
python scripts/bench_vendor_gemm.py > $O/r06_vendor_gemm_calibration.log 2>&1
# 4b. the fused feed-forward launch: against the launches it replaces, with parts of it left out, its timeline; what one SIMD sustains
python scripts/lab_ff_fused.py > $O/r06_lab_ff_fused.log 2>&1
python scripts/exp_shard_sizes.py > $O/r06_shard_sizes.log 2>&1
python scripts/lab_rowchain.py > $O/r06_lab_rowchain.log 2>&1
python scripts/lab_ff_pre.py > $O/r06_lab_ff_pre.log 2>&1
python scripts/exp_fixed_vs_variable.py > $O/r06_fixed_vs_variable.md 2>&1
# 4c. the fp16-storage engine (the mixed_precision every shipped yaml names): the same bench line, an extra measurement
python bench.py --dtype fp16 --no-cpu-baseline --no-train 2>/dev/null | cut -c1-6000 > $O/r06_bench_fp16.json.log
# 5. the other configurations
for w in bridge sthv2_512 sthv2_14 bridge_17; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-train 2>/dev/null | cut -c1-4000 >> $O/r06_bench_other_configs.json.log
done
# 6. the same step under rocprofv3 at config 4 (the 64x64 latent): where the 256 x 320 tile kernel carries the GEMM / conv class
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof4 -- python3 $R/bench.py --workload sthv2_512 --steps 10 --warmup 3 --no-cpu-baseline --no-train > $O/r06_bench_config4_under_rocprof.json.log 2>&1)
S4=$(ls $O/rocprof4/*/*_kernel_stats.csv | head -1)
python scripts/rocprof_summary.py $S4 63 > $O/r06_rocprof_summary_config4.md 2>&1
rm -rf $O/rocprof4
tail -c 600 $O/r06_bench_final.json.log; cat $O/r06_step_timeline.md | head -12; cat $O/r06_shard_sizes.log
# 6b. the fine-tuning step: grouped against per-layer weight gradients, and one steady-state step by kernel family
bash scripts/ab_train_dw.sh > $O/r06_train_dw_grouped.log 2>&1
bash scripts/run_train_profile.sh > /dev/null 2>&1
cp $R/gpurun_out/r06t/r06_train_step_profile.md $O/r06_train_step_profile.md
# 7. the whole GPU suite and the smoke test on the same tree
python -m pytest tests -m gpu -q --durations=25 > $O/r06_gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/r06_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/r06_gpu_tests.log 2>&1; echo "smoke rc $?" >> $O/r06_gpu_tests.log
tail -8 $O/r06_gpu_tests.log
