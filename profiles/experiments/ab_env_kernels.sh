# per-kernel A/B of an environment switch on one box: C2 bench, mapper loop, config 5 (rocprofv3 kernel stats)
# bash profiles/experiments/ab_env_kernels.sh "<grep pattern>" VAR valueA valueB
pat=$1; var=$2; shift 2
export R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for val in "$@"; do
  export $var=$val
  rm -rf $R/gpurun_out/abe; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abe -o x -- python3 $R/bench.py --steps 300 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "== C2 bench, $var=$val"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abe/x_results.db 2>&1 | grep -E "$pat" | cut -c1-100
  rm -rf $R/gpurun_out/abe; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abe -o x -- python3 $R/examples/mapper_loop.py > /dev/null 2>&1
  echo "== mapper loop, $var=$val"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abe/x_results.db 2>&1 | grep -E "$pat" | cut -c1-100
  rm -rf $R/gpurun_out/abe; AGS_FREEZE=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abe -o x -- python3 $R/profiles/experiments/c5_eager_steps.py > /dev/null 2>&1
  echo "== config 5, $var=$val"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abe/x_results.db 2>&1 | grep -E "$pat" | cut -c1-100
  rm -rf $R/gpurun_out/abe
done; done
