#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
bash profiles/experiments/fresh_lease_suite.sh r06 e 3 > /dev/null
tail -3 $O/r06_gpu_tests_fresh.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_z_smoke.log 2>&1; tail -3 $O/r06_z_smoke.log | cut -c1-80
python bench.py --steps 20 --warmup 5 2>$O/r06_z_bench_driver_shape.err | tail -1 > $O/r06_z_bench_driver_shape.json; cut -c1-250 $O/r06_z_bench_driver_shape.json
python examples/large_configs.py 2>&1 | tail -2 > $O/r06_z_large_configs.json; cut -c1-200 $O/r06_z_large_configs.json
