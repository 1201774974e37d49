set -x
cd $GRAFT_REPO_ROOT
python scripts/exp_flake.py --colsums 1 --cotenant 1 --iters 1000 2>&1 | grep -v Warning | tail -40 > gpurun_out/flake_a.log
python scripts/exp_flake.py --colsums 0 --cotenant 1 --iters 1000 2>&1 | tail -40 > gpurun_out/flake_b.log
python scripts/exp_flake.py --colsums 1 --cotenant 0 --iters 1000 2>&1 | tail -40 > gpurun_out/flake_c.log
python scripts/exp_flake.py --colsums 1 --cotenant 1 --trace 1 --iters 1000 2>&1 | tail -60 > gpurun_out/flake_d.log
python scripts/exp_flake.py --colsums 1 --cotenant 1 --poison 1 --iters 300 2>&1 | tail -40 > gpurun_out/flake_e.log
tail -n 8 gpurun_out/flake_*.log
