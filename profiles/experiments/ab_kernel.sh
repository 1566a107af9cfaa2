# per-kernel A/B of library builds on one box: bash profiles/experiments/ab_kernel.sh "<grep pattern>" tagA tagB ...
pat=$1; shift
export R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for tag in "$@"; do
  AGS_LIB_PATH=$R/scratch/libags_$tag.so rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abk_$tag -o x -- python3 $R/bench.py --steps 300 --no-cpu-baseline > /dev/null 2>&1
  echo "== $tag"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abk_$tag/x_results.db 2>&1 | grep -E "$pat" | cut -c1-90
done; done
