#!/bin/bash
# the library with no op_sel[1]=1 packed instruction: lab controls, both co-tenant checks, the whole GPU suite, smoke
mkdir -p gpurun_out/r03n
O=gpurun_out/r03n
timeout 120 build/lab_pkswap --seconds 2 --cotenant 0 > $O/lab_pkswap_alone.log 2>&1
timeout 200 build/lab_pkswap --seconds 3 --cotenant 1 > $O/lab_pkswap_self_cotenant.log 2>&1
timeout 900 python scripts/exp_flake_train.py --iters 2500 --graph-first 1 > $O/flake_train.log 2>&1
timeout 600 python scripts/exp_flake.py --colsums 1 --cotenant 1 --iters 1500 > $O/flake_infer.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
grep -c "wrong     0" $O/lab_pkswap_alone.log $O/lab_pkswap_self_cotenant.log
echo "== train"; grep "train\]" $O/flake_train.log; echo "== infer"; grep RESULT $O/flake_infer.log; tail -3 $O/pytest.log; tail -3 $O/smoke.log
