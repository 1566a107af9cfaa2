# per-kernel A/B of library builds on config 5 (rocprofv3 stats): bash profiles/experiments/ab_kernel_large.sh "<grep pattern>" tagA tagB ...
pat=$1; shift
export R=$GRAFT_REPO_ROOT AGS_FREEZE=1; cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  if [ "$tag" = cur ]; then unset AGS_LIB_PATH; else export AGS_LIB_PATH=$R/scratch/libags_$tag.so; fi
  rm -rf $R/gpurun_out/abkl; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abkl -o x -- python3 $R/examples/large_configs.py --only c5 > /dev/null 2>&1
  echo "== $tag"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abkl/x_results.db 2>&1 | grep -E "$pat" | cut -c1-90
done
rm -rf $R/gpurun_out/abkl
