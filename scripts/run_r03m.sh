#!/bin/bash
# lab_pkswap, all forms, next to the network as co-tenant
mkdir -p gpurun_out/r03m
O=gpurun_out/r03m
rm -f /tmp/lab_noise.stop /tmp/lab_noise.ready
python scripts/exp_flake.py --role noise --stop-file /tmp/lab_noise.stop --ready-file /tmp/lab_noise.ready > $O/lab_noise.log 2>&1 &
NP=$!
for i in $(seq 1 240); do [ -e /tmp/lab_noise.ready ] && break; sleep 0.5; done
timeout 300 build/lab_pkswap --seconds 6 --cotenant 0 > $O/lab_pkswap_network_cotenant.log 2>&1
echo stop > /tmp/lab_noise.stop; wait $NP
cat $O/lab_pkswap_network_cotenant.log
