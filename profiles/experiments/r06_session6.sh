#!/bin/bash
# final tree: the suite twice on a fresh lease, smoke, the default bench line, the mapper loop
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
bash profiles/experiments/fresh_lease_suite.sh r06 f 2 > /dev/null
tail -2 $O/r06_gpu_tests_fresh.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_z_smoke.log 2>&1; tail -2 $O/r06_z_smoke.log | cut -c1-80
python bench.py 2>$O/r06_z_bench.err | tail -1 > $O/r06_z_bench.json; cut -c1-250 $O/r06_z_bench.json
python examples/mapper_loop.py --repeat 3 2>/dev/null | cut -c150-330 > $O/r06_z_mapper_loop.txt; cat $O/r06_z_mapper_loop.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_z -o m -- python3 $R/examples/mapper_loop.py > $O/prof_z.log 2>&1
python3 $R/profiles/rocpd_stats.py $O/prof_z/m_results.db 2>&1 | head -14 | cut -c1-110 > $O/r06_z_mapper_loop_kernel_stats.md; cat $O/r06_z_mapper_loop_kernel_stats.md
rm -rf $O/prof_z
