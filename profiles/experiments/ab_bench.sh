# same-box A/B of bench.py variants: bash profiles/experiments/ab_bench.sh <runs> "<label>|<env assignments>|<bench args>" ...
# e.g. bash profiles/experiments/ab_bench.sh 2 "direct||--binning direct" "tsort||--binning tile_sort" "noearly|AGS_LIB_PATH=$PWD/scratch/libags_noearly.so|"
runs=$1; shift
for r in $(seq 1 $runs); do for spec in "$@"; do
  IFS='|' read -r label envs args <<< "$spec"
  env $envs python bench.py --no-cpu-baseline $args 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],5), d['config']['stage_ms'])"
done; done
