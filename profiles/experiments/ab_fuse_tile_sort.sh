#!/bin/bash
# A/B of AgsTuning.fuse_tile_sort on bench.py's C2 step: per-kernel rocprofv3 averages and the step time, interleaved
export R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do for f in 0 1; do
  rm -rf $R/gpurun_out/abf_$f
  AGS_FUSE_TILE_SORT=$f rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abf_$f -o x -- python3 $R/bench.py --steps 300 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "== C2 kernels, fuse_tile_sort=$f"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abf_$f/x_results.db 2>&1 | grep -E "render_fwd|render_bwd|tile_sort|ags_k_preprocess<|bwd_rows" | cut -c1-100
  rm -rf $R/gpurun_out/abf_$f
  AGS_FUSE_TILE_SORT=$f python3 $R/bench.py --steps 20 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step ms', round(d['ms_per_step'],5), 'pipelined', d.get('ms_per_step_pipelined'))"
done; done
