#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (profiles/experiments/fetch_calibration.hip): kernel durations + the two counters in
# separate passes, then the per-pattern factors.   usage: bash profiles/experiments/fetch_calibration.sh <tag>
#   -> gpurun_out/<tag>_fetch_calibration.md  (copy into profiles/)
TAG=${1:-r00}
R=${GRAFT_REPO_ROOT:-$(pwd)}
EXE=$R/gpurun_out/fetch_calibration
mkdir -p $R/gpurun_out
hipcc --offload-arch=gfx950 -O3 $R/profiles/experiments/fetch_calibration.hip -o $EXE || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/cal_k; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cal_k -o k -- $EXE > $R/gpurun_out/cal_bytes.json 2> $R/gpurun_out/cal_k.log
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/cal_$c
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/cal_$c -o p -- $EXE > /dev/null 2> $R/gpurun_out/cal_$c.log
done
cd $R
python3 profiles/experiments/fetch_calibration_summary.py gpurun_out/cal_bytes.json gpurun_out/cal_FETCH_SIZE/p_counter_collection.csv \
    gpurun_out/cal_WRITE_SIZE/p_counter_collection.csv gpurun_out/cal_k/k_kernel_stats.csv $TAG | tee gpurun_out/${TAG}_fetch_calibration.md
rm -rf gpurun_out/cal_k gpurun_out/cal_FETCH_SIZE gpurun_out/cal_WRITE_SIZE $EXE
