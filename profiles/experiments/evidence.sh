#!/bin/bash
# One round's evidence batch on the GPU box: bash profiles/experiments/evidence.sh <tag>
# -> gpurun_out/<tag>_{bench,large_configs,mapper_loop,planner_views,dropin}.json, <tag>_kernel_stats.md,
#    <tag>_mapper_loop_kernel_stats.md, pmc_hbm_bytes.json, sq_counters.json, <tag>_sq_counters.md (copy into profiles/).
TAG=${1:-r00}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
python bench.py 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-200 gpurun_out/${TAG}_bench.json
python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 > gpurun_out/${TAG}_bench_driver_shape.json; cut -c1-200 gpurun_out/${TAG}_bench_driver_shape.json
python examples/large_configs.py 2>&1 | tail -2 > gpurun_out/${TAG}_large_configs.json; cut -c1-170 gpurun_out/${TAG}_large_configs.json
# (twice: the first process that runs the loop on a box pays one-time costs outside the library - 0.9 s against 0.67 s)
python examples/mapper_loop.py > /dev/null 2>&1
python examples/mapper_loop.py 2>&1 | tail -1 > gpurun_out/${TAG}_mapper_loop.json; cut -c150-330 gpurun_out/${TAG}_mapper_loop.json
# BASELINE.md lists configuration 3 at 1200x680; the reference's simulator renders 512x512 (habitat.yaml): both for the record
python examples/mapper_loop.py --size 680 1200 2>&1 | tail -1 > gpurun_out/${TAG}_mapper_loop_1200x680.json; cut -c150-330 gpurun_out/${TAG}_mapper_loop_1200x680.json
python examples/planner_views.py 2>&1 | tail -1 > gpurun_out/${TAG}_planner_views.json; cat gpurun_out/${TAG}_planner_views.json
python examples/dropin_path.py 2>&1 | tail -2 > gpurun_out/${TAG}_dropin.json; cut -c1-300 gpurun_out/${TAG}_dropin.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -o k -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${TAG}.log 2>&1
python3 $R/profiles/rocpd_stats.py $R/gpurun_out/prof_${TAG}/k_results.db 2>&1 | head -8 > $R/gpurun_out/${TAG}_kernel_stats.md; cat $R/gpurun_out/${TAG}_kernel_stats.md
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}m -o m -- python3 $R/examples/mapper_loop.py > $R/gpurun_out/prof_${TAG}m.log 2>&1
python3 $R/profiles/rocpd_stats.py $R/gpurun_out/prof_${TAG}m/m_results.db 2>&1 | head -16 | cut -c1-110 > $R/gpurun_out/${TAG}_mapper_loop_kernel_stats.md; head -4 $R/gpurun_out/${TAG}_mapper_loop_kernel_stats.md
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}l -o l -- python3 $R/examples/large_configs.py --only c5 > $R/gpurun_out/prof_${TAG}l.log 2>&1
python3 $R/profiles/rocpd_stats.py $R/gpurun_out/prof_${TAG}l/l_results.db 2>&1 | head -10 | cut -c1-110 > $R/gpurun_out/${TAG}_c5_kernel_stats.md; head -8 $R/gpurun_out/${TAG}_c5_kernel_stats.md
rm -rf $R/gpurun_out/prof_${TAG} $R/gpurun_out/prof_${TAG}m $R/gpurun_out/prof_${TAG}l
cd $R
bash profiles/experiments/pmc_run.sh $TAG
bash profiles/experiments/pmc_sq.sh $TAG
