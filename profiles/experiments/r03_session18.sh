#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_distributed.py tests/test_gpu_pipeline.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['ms_per_step'], d['config']['stage_ms'])"
cd /tmp && export TMPDIR=/tmp AGS_FREEZE=1
rm -rf $R/gpurun_out/abk; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abk -o x -- python3 $R/examples/large_configs.py > /dev/null 2>&1
python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abk/x_results.db 2>&1 | grep "rows" | cut -c1-100
rm -rf $R/gpurun_out/abk
