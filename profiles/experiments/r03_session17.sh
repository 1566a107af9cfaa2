#!/bin/bash
# coverage runs: the whole GPU suite with the cull-first kernel forced at every size, and with a status read-back per module call
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
AGS_PRE_CULL_MIN_N=0 timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -k "not cull_first" 2>&1 | tail -6
AGS_DROPIN_STATUS=always timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_consumers.py tests/test_gpu_golden.py -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -4
