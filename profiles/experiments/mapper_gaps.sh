#!/bin/bash
# Where the GPU idles in the mapper loop (config 3): rocprofv3 kernel trace of examples/mapper_loop.py, kernel stats, the
# idle gaps by (ended -> started) kernel pair from the first timed keyframe on, and the busy fraction of the loop.
# usage: bash profiles/experiments/mapper_gaps.sh <tag> [extra mapper_loop.py args]
TAG=${1:-r00}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python3 $R/examples/mapper_loop.py "$@" > /dev/null 2>&1      # the first process on a box pays one-time costs
python3 $R/examples/mapper_loop.py "$@" 2>&1 | tail -1 > $R/gpurun_out/${TAG}_mapper_loop.json
rm -rf $R/gpurun_out/mg; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/mg -o m -- python3 $R/examples/mapper_loop.py "$@" > $R/gpurun_out/${TAG}_mapper_prof.log 2>&1
python3 $R/profiles/rocpd_stats.py $R/gpurun_out/mg/m_results.db 2>&1 | head -24 | cut -c1-120 > $R/gpurun_out/${TAG}_mapper_loop_kernel_stats.md
python3 $R/profiles/rocpd_gaps.py $R/gpurun_out/mg/m_results.db 5 40 ags_k_bilateral@-50 > $R/gpurun_out/${TAG}_mapper_loop_gaps.md
python3 $R/profiles/experiments/mapper_busy.py $R/gpurun_out/mg/m_results.db $TAG > $R/gpurun_out/${TAG}_c3_kernels_busy.json
rm -rf $R/gpurun_out/mg
cut -c1-400 $R/gpurun_out/${TAG}_mapper_loop.json; cat $R/gpurun_out/${TAG}_c3_kernels_busy.json; head -44 $R/gpurun_out/${TAG}_mapper_loop_gaps.md | cut -c1-150; head -24 $R/gpurun_out/${TAG}_mapper_loop_kernel_stats.md
