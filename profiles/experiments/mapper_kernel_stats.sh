#!/bin/bash
# rocprofv3 per-kernel averages of the mapper loop (config 3), top lines only.  usage: bash profiles/experiments/mapper_kernel_stats.sh <tag> [lines]
TAG=${1:-r00}; N=${2:-16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/mk; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/mk -o m -- python3 $R/examples/mapper_loop.py > /dev/null 2>&1
python3 $R/profiles/rocpd_stats.py $R/gpurun_out/mk/m_results.db 2>&1 | head -$((N + 2)) | cut -c1-120 > $R/gpurun_out/${TAG}_mapper_loop_kernel_stats.md
rm -rf $R/gpurun_out/mk
cat $R/gpurun_out/${TAG}_mapper_loop_kernel_stats.md
