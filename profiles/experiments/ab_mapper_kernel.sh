# per-kernel A/B of library builds on the mapper loop (config 3) and on bench.py's C2 step, one box:
# bash profiles/experiments/ab_mapper_kernel.sh "<grep pattern>" tagA tagB ...   (builds: scratch/libags_<tag>.so)
pat=$1; shift
export R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for tag in "$@"; do
  rm -rf $R/gpurun_out/abm_$tag
  AGS_LIB_PATH=$R/scratch/libags_$tag.so rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abm_$tag -o x -- python3 $R/examples/mapper_loop.py > /dev/null 2>&1
  echo "== mapper loop, $tag"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abm_$tag/x_results.db 2>&1 | grep -E "$pat" | cut -c1-100
  AGS_LIB_PATH=$R/scratch/libags_$tag.so python3 $R/examples/mapper_loop.py | cut -c230-290
  rm -rf $R/gpurun_out/abm_$tag
  AGS_LIB_PATH=$R/scratch/libags_$tag.so rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abm_$tag -o x -- python3 $R/bench.py --steps 300 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "== C2 bench, $tag"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abm_$tag/x_results.db 2>&1 | grep -E "$pat" | cut -c1-100
  rm -rf $R/gpurun_out/abm_$tag
done; done
