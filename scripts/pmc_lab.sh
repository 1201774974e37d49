#!/bin/bash
# PMC passes over one lab binary (separate passes: --pmc only, no trace domains).  usage: scripts/pmc_lab.sh <tag> <cmd...>
# Results: gpurun_out/<tag>/pass_*/**/*counter_collection.csv, summarised by scripts/pmc_lab_summary.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $out/pass_$i -o p$i --output-format csv -- "$@" > $out/pass_$i.log 2>&1
done
ls -R $out | head -40
