#!/bin/bash
# round 3, GPU session 3: the native drop-in path (tests + timings), the surfel-order probe
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export AGS_PARITY_LOG=$R/gpurun_out/r03_parity_log.jsonl; rm -f $AGS_PARITY_LOG
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r03_pytest.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r03_pytest.log
unset AGS_PARITY_LOG
python profiles/experiments/prof_dropin.py 680 1200 1 2>&1 | grep -v "^$" | head -40 > gpurun_out/r03_prof_dropin_c2.txt
python profiles/experiments/prof_dropin.py 512 512 8 2>&1 | grep -v "^$" | head -40 > gpurun_out/r03_prof_dropin_512.txt
head -4 gpurun_out/r03_prof_dropin_c2.txt | cut -c1-300; head -4 gpurun_out/r03_prof_dropin_512.txt | cut -c1-300
timeout 900 python profiles/experiments/morton_probe.py 2>&1 | tail -8
