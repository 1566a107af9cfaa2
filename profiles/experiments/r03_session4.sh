#!/bin/bash
# round 3, GPU session 4: bench line, drop-in example, HIP API trace of the drop-in loop (synchronisations per call)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r03_b_bench.json 2> gpurun_out/r03_b_bench.err; echo "bench rc $?"; cut -c1-300 gpurun_out/r03_b_bench.json; tail -3 gpurun_out/r03_b_bench.err
timeout 600 python examples/dropin_path.py 2>&1 | tail -2 > gpurun_out/r03_b_dropin.json; cut -c1-420 gpurun_out/r03_b_dropin.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/hipt; timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d $R/gpurun_out/hipt -o h -- python3 $R/profiles/experiments/prof_dropin.py 512 512 8 > $R/gpurun_out/r03_hiptrace.log 2>&1
ls $R/gpurun_out/hipt $R/gpurun_out/hipt/* | head -20
f=$(ls $R/gpurun_out/hipt/*hip_api_stats.csv $R/gpurun_out/hipt/*/*hip_api_stats.csv 2>/dev/null | head -1)
{ echo "# r03-b: HIP API calls of profiles/experiments/prof_dropin.py 512 512 8 (rocprofv3 --hip-trace --stats): 93 iterations x 8 views = 744 module calls + backward"; head -30 "$f"; tail -3 $R/gpurun_out/r03_hiptrace.log; } > $R/gpurun_out/r03_b_dropin_hip_api_stats.md
cat $R/gpurun_out/r03_b_dropin_hip_api_stats.md | cut -c1-200
rm -rf $R/gpurun_out/hipt
