#!/bin/bash
# final: evidence set, then the whole GPU suite and smoke on the same library
bash scripts/run_evidence_r03.sh > gpurun_out/evidence_stdout.log 2>&1
tail -c 1500 gpurun_out/evidence_stdout.log
mkdir -p gpurun_out/r03x
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r03x/pytest.log 2>&1; tail -3 gpurun_out/r03x/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03x/smoke.log 2>&1; tail -2 gpurun_out/r03x/smoke.log
