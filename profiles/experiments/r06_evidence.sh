#!/bin/bash
# Round 6's evidence batch of the final tree, in one gpurun call on a FRESH lease:  bash profiles/experiments/r06_evidence.sh <tag> <lease>
TAG=${1:-r06_z}; LEASE=${2:-d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
bash profiles/experiments/fresh_lease_suite.sh r06 $LEASE 3 > /dev/null; tail -3 gpurun_out/r06_gpu_tests_fresh.log
export AGS_PARITY_LOG=$R/gpurun_out/${TAG}_parity_log.jsonl; rm -f $AGS_PARITY_LOG
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/${TAG}_gpu_tests.log; cat gpurun_out/${TAG}_gpu_tests.log
unset AGS_PARITY_LOG
python profiles/experiments/parity_summary.py gpurun_out/${TAG}_parity_log.jsonl > gpurun_out/${TAG}_parity_margins.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; tail -17 gpurun_out/${TAG}_smoke.log | cut -c1-60
# bench lines, examples, kernel stats of C2 / mapper loop / C5 (evidence.sh also runs the FETCH_SIZE / WRITE_SIZE and SQ passes of C2)
bash profiles/experiments/evidence.sh $TAG
bash profiles/experiments/c5_counters.sh $TAG
# memory-side request counters (bracketed reads, exact writes): C2's eager bench step and configuration 5's steady-state steps
bash profiles/experiments/pmc_exact.sh $TAG c2 $R/gpurun_out/pmc_hbm_bytes.json -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras --eager > /dev/null 2>&1; cat gpurun_out/${TAG}_c2_pmc_exact.md
AGS_FREEZE=1 bash profiles/experiments/pmc_exact.sh $TAG c5 $R/gpurun_out/pmc5_hbm_bytes.json -- python3 $R/profiles/experiments/c5_eager_steps.py > /dev/null 2>&1; cat gpurun_out/${TAG}_c5_pmc_exact.md
bash profiles/experiments/mapper_gaps.sh $TAG
python examples/mission_loop.py 2>&1 | tail -1 > gpurun_out/${TAG}_mission_loop.json; cut -c1-300 gpurun_out/${TAG}_mission_loop.json
python profiles/experiments/mapper_phases_r05.py > gpurun_out/${TAG}_mapper_phases.jsonl 2>&1; cat gpurun_out/${TAG}_mapper_phases.jsonl | cut -c1-600
