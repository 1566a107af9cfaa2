#!/bin/bash
# round-6 session 1: the failing one-rank RCCL test alone (full output), the suite with durations, the reduce-form error
# table, counter calibration, one bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -k rccl_collectives 2>&1 | tail -60 > $O/r06_s1_rccl_alone.log
python -m pytest tests -m gpu -x -q --durations=0 2>&1 | tail -260 > $O/r06_s1_suite.log
tail -3 $O/r06_s1_suite.log
timeout 1500 python profiles/experiments/bwd_reduce_error.py $O/r06_bwd_reduce_error_table.md > $O/r06_bwd_reduce_error.jsonl 2> $O/r06_bwd_reduce_error.err
tail -2 $O/r06_bwd_reduce_error.err
bash profiles/experiments/fetch_calibration.sh r06 > /dev/null 2>&1
cat $O/r06_fetch_calibration.md
python bench.py > $O/r06_s1_bench.json 2> $O/r06_s1_bench.err; tail -c 600 $O/r06_s1_bench.json
