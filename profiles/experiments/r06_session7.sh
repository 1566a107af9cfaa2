#!/bin/bash
# the batched per-Gaussian stage with row loads shared across views (AgsTuning.view_group): test, A/B on the frozen mapper
# batch (per kernel), the mapper loop, the planner batch
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q -k "shares_row_loads or view_batch or planner_shaped or batched_iteration or frame_store or seen_flags" 2>&1 | tail -30 > $O/r06_s7_tests.log; tail -3 $O/r06_s7_tests.log
python profiles/experiments/mapper_frozen_steps.py make /tmp/frozen_map.pt 2>&1 | tail -1
for vg in 1 2 4 6 11 0; do echo "== frozen mapper batch AGS_VIEW_GROUP=$vg: $(AGS_VIEW_GROUP=$vg python profiles/experiments/mapper_frozen_steps.py run /tmp/frozen_map.pt 2>/dev/null | grep ms/iteration | tr '\n' ';')"; done > $O/r06_s7_view_group_ab.txt
cat $O/r06_s7_view_group_ab.txt
cd /tmp && export TMPDIR=/tmp
for vg in 1 4; do
  rm -rf $O/abv; AGS_VIEW_GROUP=$vg rocprofv3 --kernel-trace --stats -d $O/abv -o x -- python3 $R/profiles/experiments/mapper_frozen_steps.py run /tmp/frozen_map.pt > /dev/null 2>&1
  echo "== kernels, AGS_VIEW_GROUP=$vg"; python3 $R/profiles/rocpd_stats.py $O/abv/x_results.db 2>&1 | grep -E "preprocess|tile_sort|render_fwd" | cut -c1-110
  rm -rf $O/abv
done >> $O/r06_s7_view_group_ab.txt
cd $R
for vg in 1 0 1 0; do echo "== mapper loop AGS_VIEW_GROUP=$vg: $(AGS_VIEW_GROUP=$vg python examples/mapper_loop.py --repeat 3 2>/dev/null | python -c "
import sys, json
print([json.loads(l)['seconds'] for l in sys.stdin if l.startswith('{')])")"; done >> $O/r06_s7_view_group_ab.txt
for vg in 1 0; do echo "== planner views AGS_VIEW_GROUP=$vg: $(AGS_VIEW_GROUP=$vg python examples/planner_views.py 2>/dev/null | tail -1 | cut -c1-330)"; done >> $O/r06_s7_view_group_ab.txt
tail -12 $O/r06_s7_view_group_ab.txt
