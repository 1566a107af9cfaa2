#!/bin/bash
# The GPU suite as the FIRST command of a fresh lease, N times in a row; one summary line per run (and the failure's
# output, if any) appended to gpurun_out/<tag>_gpu_tests_fresh.log.   usage: bash fresh_lease_suite.sh <tag> <lease> <runs>
TAG=${1:-r00}; LEASE=${2:-a}; N=${3:-4}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
for i in $(seq 1 $N); do
  python -m pytest tests -m gpu -x -q > $O/suite_run.log 2>&1
  rc=$?
  echo "lease $LEASE run $i (host $(hostname), $(date -u +%H:%M:%S)): rc=$rc $(tail -1 $O/suite_run.log)" >> $O/${TAG}_gpu_tests_fresh.log
  if [ $rc -ne 0 ]; then tail -60 $O/suite_run.log >> $O/${TAG}_gpu_tests_fresh.log; fi
done
rm -f $O/suite_run.log
cat $O/${TAG}_gpu_tests_fresh.log
