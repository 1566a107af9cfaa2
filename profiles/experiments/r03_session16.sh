#!/bin/bash
# round 3, GPU session 16: whole GPU suite + the round's final evidence batch (tag r03_d)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export AGS_PARITY_LOG=$R/gpurun_out/r03_parity_log.jsonl; rm -f $AGS_PARITY_LOG
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r03_pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r03_pytest.log
unset AGS_PARITY_LOG
bash profiles/experiments/evidence.sh r03_d 2>&1 | tail -50
