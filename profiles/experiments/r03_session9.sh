#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cull_kernel.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for thr in 1048576 1073741824; do
  echo "== AGS_PRE_CULL_MIN_N=$thr"
  AGS_PRE_CULL_MIN_N=$thr python examples/large_configs.py 2>&1 | tail -2 | cut -c1-130
done
cd /tmp && export TMPDIR=/tmp AGS_FREEZE=1
for thr in 1048576 1073741824; do
  rm -rf $R/gpurun_out/abk; AGS_PRE_CULL_MIN_N=$thr rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abk -o x -- python3 $R/examples/large_configs.py > /dev/null 2>&1
  echo "== AGS_PRE_CULL_MIN_N=$thr (frozen scenes, c4 + c5)"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abk/x_results.db 2>&1 | grep preprocess | cut -c1-100
done
rm -rf $R/gpurun_out/abk
