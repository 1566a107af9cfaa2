cd /tmp && export TMPDIR=/tmp
for tag in noatomic nocount noemit; do
  export AGS_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libags_$tag.so
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/exp_$tag -o x -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/exp_$tag.log 2>&1
  echo "== $tag"; python3 $GRAFT_REPO_ROOT/profiles/rocpd_stats.py $GRAFT_REPO_ROOT/gpurun_out/exp_$tag/x_results.db 2>&1 | grep -E "preprocess<|bucket|scan|tile_sort" | cut -c1-90
done
