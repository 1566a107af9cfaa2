#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
bash profiles/experiments/fresh_lease_suite.sh r06 c 3 > /dev/null
tail -4 $O/r06_gpu_tests_fresh.log
hipcc --offload-arch=gfx950 -O3 profiles/experiments/fetch_calibration.hip -o $O/fetch_calibration
bash profiles/experiments/pmc_exact.sh r06 calibration - -- $O/fetch_calibration > /dev/null 2>&1; cat $O/r06_calibration_pmc_exact.md
rm -f $O/fetch_calibration
bash profiles/experiments/pmc_exact.sh r06 c2 $O/pmc_hbm_bytes.json -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras --eager > /dev/null 2>&1; cat $O/r06_c2_pmc_exact.md
AGS_FREEZE=1 bash profiles/experiments/pmc_exact.sh r06 c5 $O/pmc5_hbm_bytes.json -- python3 $R/profiles/experiments/c5_eager_steps.py > /dev/null 2>&1; cat $O/r06_c5_pmc_exact.md
python profiles/experiments/morton_probe.py 2>&1 | grep -v Warning | tail -8 | cut -c1-400
cp $O/r03_morton_probe.json $O/r06_morton_probe.json 2>/dev/null
