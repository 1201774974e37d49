# the fine-tuning step with the weight gradients (and the LayerNorms' d gamma / d beta finals) one launch per layer / grouped at the end of
# each backward walk, and the rows of contraction a workgroup of the grouped launch takes:
#   bash scripts/ab_train_dw.sh > profiles/r06_train_dw_grouped.log
line() { python scripts/bench_train.py 10 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print({k: round(j[k], 3) for k in ('fwd_bwd_ms', 'optimizer_ms', 'ms_per_step', 'loss_first', 'loss_last')})"; }
echo "SEER_DW_GROUPED=0:"; SEER_DW_GROUPED=0 line
for r in 2048 16384; do echo "grouped, SEER_TN_GROUP_ROWS=$r:"; SEER_TN_GROUP_ROWS=$r line; done
echo "SEER_DW_GROUPED=0:"; SEER_DW_GROUPED=0 line
echo "grouped (default):"; line
