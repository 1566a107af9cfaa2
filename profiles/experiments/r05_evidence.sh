#!/bin/bash
# Round 5's evidence batch of the final tree, in one gpurun call:  bash profiles/experiments/r05_evidence.sh <tag>
TAG=${1:-r05_h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
export AGS_PARITY_LOG=$R/gpurun_out/${TAG}_parity_log.jsonl; rm -f $AGS_PARITY_LOG
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/${TAG}_gpu_tests.log; cat gpurun_out/${TAG}_gpu_tests.log
unset AGS_PARITY_LOG
python profiles/experiments/parity_summary.py gpurun_out/${TAG}_parity_log.jsonl > gpurun_out/${TAG}_parity_margins.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
bash profiles/experiments/evidence.sh $TAG
bash profiles/experiments/c5_counters.sh $TAG
bash profiles/experiments/mapper_gaps.sh $TAG
bash profiles/experiments/pmc_preprocess_c2.sh $TAG
python examples/mission_loop.py 2>&1 | tail -1 > gpurun_out/${TAG}_mission_loop.json; cut -c1-300 gpurun_out/${TAG}_mission_loop.json
python profiles/experiments/mapper_phases_r05.py > gpurun_out/${TAG}_mapper_phases.jsonl 2>&1; cat gpurun_out/${TAG}_mapper_phases.jsonl
