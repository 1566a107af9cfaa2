# bash scratch/ab3.sh tagA tagB: C2 x3, c4/c5, mapper bwd kernel time
for r in 1 2 3; do for tag in "$@"; do
  AGS_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libags_$tag.so python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],5), d['config']['stage_ms'])"
done; done
for tag in "$@"; do export AGS_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libags_$tag.so; echo "== $tag"
python examples/large_configs.py --steps 20 2>&1 | tail -2 | cut -c1-140
done
cd /tmp && export TMPDIR=/tmp R=$GRAFT_REPO_ROOT
for tag in "$@"; do export AGS_LIB_PATH=$R/scratch/libags_$tag.so
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_h$tag -o m -- python3 $R/examples/mapper_loop.py > /dev/null 2>&1
python3 $R/profiles/rocpd_stats.py $R/gpurun_out/prof_h$tag/m_results.db 2>&1 | sed -n 3,3p | cut -c1-110; rm -rf $R/gpurun_out/prof_h$tag
done
