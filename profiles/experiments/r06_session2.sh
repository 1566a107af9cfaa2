#!/bin/bash
# round-6 session 2: new tests, smoke sweep, A/B of the reduce forms on config 5 and the mapper loop, loss-epilogue and
# host-pose A/B on the mapper loop, atomic counters of config 5
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q -k "epilogue or host_pose or readers_of_the_map or bf16_split or four_ranks or single_rank_contract" 2>&1 | tail -40 > $O/r06_s2_new_tests.log
tail -3 $O/r06_s2_new_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_s2_smoke.log 2>&1; tail -22 $O/r06_s2_smoke.log
for m in f32 bf16x3 bf16; do echo "== c5 AGS_BWD_REDUCE=$m"; AGS_FREEZE=1 AGS_BWD_REDUCE=$m python profiles/experiments/c5_eager_steps.py 2>&1 | grep ms/step; done > $O/r06_s2_c5_reduce_ab.txt
cat $O/r06_s2_c5_reduce_ab.txt
for cfg in "AGS_BWD_REDUCE=f32" "AGS_BWD_REDUCE=bf16x3" "AGS_BWD_REDUCE=bf16x3 AGS_FUSE_LOSS_STAGE1=0" "AGS_BWD_REDUCE=f32 AGS_FUSE_LOSS_STAGE1=0"; do
  echo "== mapper loop 512x512: $cfg"; env $cfg python examples/mapper_loop.py --repeat 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['seconds'], d['final_surfels'], d['device_mallocs'], d['overflow_retries'])"
done > $O/r06_s2_mapper_ab.txt
echo "== mapper loop 512x512: device pose (default reduce, fused stage 1)" >> $O/r06_s2_mapper_ab.txt
python examples/mapper_loop.py --repeat 4 --device-pose 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['seconds'], d['final_surfels'], d['device_mallocs'], d['overflow_retries'])" >> $O/r06_s2_mapper_ab.txt
cat $O/r06_s2_mapper_ab.txt
bash profiles/experiments/c5_atomic_counters.sh r06 > /dev/null 2>&1; cat $O/r06_c5_atomic_counters.md | cut -c1-600
