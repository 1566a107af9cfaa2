# A/B two builds of the library on the same box: bash profiles/experiments/ab.sh old new [runs]
for r in $(seq 1 ${3:-2}); do for tag in $1 $2; do
  AGS_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libags_$tag.so python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],5), d['config']['stage_ms'])"
done; done
