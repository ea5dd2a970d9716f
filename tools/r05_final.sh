#!/bin/bash
# round 5: the evidence set of the final tree (tools/evidence_round.sh) + the default bench line as the driver runs it
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
bash tools/evidence_round.sh ${1:-r05_a} > gpurun_out/${1:-r05_a}_evidence.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/${1:-r05_a}_bench_driver.json 2> gpurun_out/${1:-r05_a}_bench_driver.err
echo done
