#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_golden.py tests/test_gpu_densify.py tests/test_gpu_fused_loss.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
python examples/mapper_loop.py > /dev/null 2>&1
python examples/mapper_loop.py 2>&1 | tail -1 | cut -c150-420
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/abk; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abk -o x -- python3 $R/examples/mapper_loop.py > /dev/null 2>&1
python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abk/x_results.db 2>&1 | head -12 | cut -c1-100
rm -rf $R/gpurun_out/abk
