#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
python -m pytest tests -q -m gpu > gpurun_out/r05_a_tests.log 2>&1; tail -3 gpurun_out/r05_a_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_a_smoke.log 2>&1; tail -1 gpurun_out/r05_a_smoke.log
(grep -E "passed|failed" gpurun_out/r05_a_tests.log; tail -1 gpurun_out/r05_a_smoke.log) > gpurun_out/r05_a_gpu_tests.txt
echo done
