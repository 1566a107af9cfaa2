#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
bash profiles/experiments/fresh_lease_suite.sh r06 g 2 > /dev/null
tail -2 $O/r06_gpu_tests_fresh.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_z_smoke.log 2>&1; tail -2 $O/r06_z_smoke.log | cut -c1-80
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r06_z_bench_driver_shape.json; cut -c1-330 $O/r06_z_bench_driver_shape.json
