#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
bash profiles/experiments/fresh_lease_suite.sh r06 b 4 > /dev/null
tail -6 $O/r06_gpu_tests_fresh.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_s3_smoke.log 2>&1; tail -18 $O/r06_s3_smoke.log
bash profiles/experiments/fetch_calibration.sh r06 > /dev/null 2>&1; cat $O/r06_fetch_calibration.md
python profiles/experiments/mapper_mallocs.py > $O/r06_mapper_mallocs.jsonl 2>/dev/null; cut -c1-1500 $O/r06_mapper_mallocs.jsonl
