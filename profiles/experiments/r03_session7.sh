#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp AGS_FREEZE=1
rm -rf $R/gpurun_out/abk; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abk -o x -- python3 $R/examples/large_configs.py --only c5 > /dev/null 2>&1
echo "== c5 (frozen scene)"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abk/x_results.db 2>&1 | head -7 | cut -c1-100
rm -rf $R/gpurun_out/abk; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abk -o x -- python3 $R/examples/mapper_loop.py > /dev/null 2>&1
echo "== mapper loop"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abk/x_results.db 2>&1 | head -9 | cut -c1-100
rm -rf $R/gpurun_out/abk
