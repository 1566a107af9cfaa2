#!/bin/bash
# The kernel sequence of one keyframe of the mapper loop (config 3), to see which small launches make up its set-up phases.
# usage: bash profiles/experiments/mapper_sequence.sh <tag> <keyframe (negative: from the end)>
TAG=${1:-r00}; K=${2:--10}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ms; rocprofv3 --kernel-trace -d $R/gpurun_out/ms -o m -- python3 $R/examples/mapper_loop.py > /dev/null 2>&1
python3 $R/profiles/rocpd_sequence.py $R/gpurun_out/ms/m_results.db ags_k_bilateral $K > $R/gpurun_out/${TAG}_mapper_keyframe_sequence.md
rm -rf $R/gpurun_out/ms
grep -v "render_\|ags_k_loss\|rows_multi\|stage_frames\|tile_sort\|ags_k_preprocess\|weighted_topk" $R/gpurun_out/${TAG}_mapper_keyframe_sequence.md | cut -c1-140
