#!/bin/bash
# round 3, GPU session 5: two-quadrant matrix-core backward: parity, then A/B on config 5 and the C4 share
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_golden.py -m gpu -q -x -k "two_quadrants" -p no:cacheprovider 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "c5 or overfull" -p no:cacheprovider 2>&1 | tail -5
for v in 1 0; do
  echo "== AGS_BWD_MFMA2=$v"
  AGS_BWD_MFMA2=$v python examples/large_configs.py --only c5 2>&1 | tail -1 | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp AGS_FREEZE=1
for v in 1 0; do
  rm -rf $R/gpurun_out/abk; AGS_BWD_MFMA2=$v rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abk -o x -- python3 $R/examples/large_configs.py --only c5 > /dev/null 2>&1
  echo "== AGS_BWD_MFMA2=$v (frozen scene)"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abk/x_results.db 2>&1 | head -8 | cut -c1-100
done
rm -rf $R/gpurun_out/abk
