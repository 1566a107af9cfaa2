#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
bash profiles/experiments/fresh_lease_suite.sh r06 h 2 > /dev/null
tail -2 $O/r06_gpu_tests_fresh.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | cut -c1-80
python bench.py > $O/r06_z_final_bench.out 2> $O/r06_z_final_bench.err; echo "stdout lines: $(wc -l < $O/r06_z_final_bench.out)"; cut -c1-260 $O/r06_z_final_bench.out
