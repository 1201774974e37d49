cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/pytest.log
python scripts/exp_flake_train.py --iters 150 2>&1 | grep -v amdgpu | tail -12 > $O/flake_train.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > $O/smoke.log
cat $O/pytest.log $O/flake_train.log $O/smoke.log
