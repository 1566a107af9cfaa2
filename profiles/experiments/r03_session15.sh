#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_distributed.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0")
r = bench.measure_config("c4", 1_500_000, 680, 1200, 4, "room0", 20, dev)
print("c4", r["ms_per_step"], r["ms_per_step_min"], r["stage_ms_per_view"], r["member_rows"])
PY
python examples/mapper_loop.py > /dev/null 2>&1
python examples/mapper_loop.py 2>&1 | tail -1 | cut -c150-330
