#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_pipeline.py tests/test_gpu_golden.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -12
python examples/large_configs.py --only c4 2>&1 | tail -1 | cut -c1-200
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from active_gs_amd.trainer import SurfelTrainer
dev = torch.device("cuda:0")
for multi in (True, False):
    SurfelTrainer.MULTI_VIEW_ROWS = multi
    r = bench.measure_config("c4", 1_500_000, 680, 1200, 4, "room0", 20, dev)
    print("MULTI_VIEW_ROWS", multi, r["ms_per_step"], r["ms_per_step_min"], r["stage_ms_per_view"])
PY
