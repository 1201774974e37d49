#!/bin/bash
# (1) which packed-fp32 forms fail next to a co-tenant (lab_pkswap), (2) the training and inference co-tenant checks on the
# library without any half-swapped packed read, (3) the kernel tests the change touches
mkdir -p gpurun_out/r03l
O=gpurun_out/r03l
timeout 300 build/lab_pkswap --seconds 10 --cotenant 1 > $O/lab_pkswap_cotenant.log 2>&1
timeout 120 build/lab_pkswap --seconds 3 --cotenant 0 > $O/lab_pkswap_alone.log 2>&1
# the network as the co-tenant of the lab
rm -f /tmp/lab_noise.stop /tmp/lab_noise.ready
python scripts/exp_flake.py --role noise --stop-file /tmp/lab_noise.stop --ready-file /tmp/lab_noise.ready > $O/lab_noise.log 2>&1 &
NP=$!
for i in $(seq 1 240); do [ -e /tmp/lab_noise.ready ] && break; sleep 0.5; done
timeout 200 build/lab_pkswap --seconds 8 --cotenant 0 > $O/lab_pkswap_network_cotenant.log 2>&1
echo stop > /tmp/lab_noise.stop; wait $NP
timeout 900 python scripts/exp_flake_train.py --iters 2500 --graph-first 1 > $O/flake_train.log 2>&1
timeout 600 python scripts/exp_flake.py --colsums 1 --cotenant 1 --iters 1500 > $O/flake_infer.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "cotenant or layernorm or rotary or trainer or train" > $O/pytest_subset.log 2>&1
for f in lab_pkswap_cotenant lab_pkswap_alone lab_pkswap_network_cotenant; do echo "== $f"; cat $O/$f.log; done
echo "== train"; grep "train\]" $O/flake_train.log; echo "== infer"; grep RESULT $O/flake_infer.log; tail -3 $O/pytest_subset.log
