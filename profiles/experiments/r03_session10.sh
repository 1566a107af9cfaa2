#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp AGS_FREEZE=1
for t in cur l2atom; do
  if [ $t = cur ]; then unset AGS_LIB_PATH; else export AGS_LIB_PATH=$R/scratch/libags_$t.so; fi
  rm -rf $R/gpurun_out/abk; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abk -o x -- python3 $R/examples/large_configs.py --only c5 > /dev/null 2>&1
  echo "== $t (frozen scene c5)"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abk/x_results.db 2>&1 | grep "preprocess\|tile_sort" | cut -c1-100
done
rm -rf $R/gpurun_out/abk
