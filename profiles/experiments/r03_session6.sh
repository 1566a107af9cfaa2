#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export AGS_LIB_PATH=$R/scratch/libags_m2s32.so
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "c5 or overfull" -p no:cacheprovider 2>&1 | tail -3
unset AGS_LIB_PATH
cd /tmp && export TMPDIR=/tmp AGS_FREEZE=1
for t in m2s32 m2s16; do
  export AGS_LIB_PATH=$R/scratch/libags_$t.so
  rm -rf $R/gpurun_out/abk; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abk -o x -- python3 $R/examples/large_configs.py --only c5 > /dev/null 2>&1
  echo "== $t (frozen scene)"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abk/x_results.db 2>&1 | grep render_bwd | cut -c1-100
done
rm -rf $R/gpurun_out/abk
