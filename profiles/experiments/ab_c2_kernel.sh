# per-kernel A/B of library builds on bench.py's C2 step only, three interleaved repetitions
# bash profiles/experiments/ab_c2_kernel.sh "<grep pattern>" tagA tagB ...
pat=$1; shift
export R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do for tag in "$@"; do
  rm -rf $R/gpurun_out/abc_$tag
  AGS_LIB_PATH=$R/scratch/libags_$tag.so rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abc_$tag -o x -- python3 $R/bench.py --steps 300 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "== C2 bench, $tag"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abc_$tag/x_results.db 2>&1 | grep -E "$pat" | cut -c1-100
  rm -rf $R/gpurun_out/abc_$tag
done; done
