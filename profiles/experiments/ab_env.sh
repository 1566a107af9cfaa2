# same-box A/B of an env switch: bash scratch/ab_env.sh VAR a b [runs]
VAR=$1; A=$2; B=$3; RUNS=${4:-3}
for i in $(seq $RUNS); do for v in $A $B; do
  export $VAR=$v
  python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],5), {k:round(x,4) for k,x in d['config']['stage_ms'].items()})"
done; done
