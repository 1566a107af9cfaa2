# same-box A/B of two checkouts: bash profiles/experiments/ab_worktree.sh scratch/wt_i . [runs]
export R=$GRAFT_REPO_ROOT
for r in $(seq 1 ${3:-2}); do for d in $1 $2; do
  (cd $R/$d && python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', round(d['ms_per_step'],5), d['config']['stage_ms'])")
done; done
cd /tmp && export TMPDIR=/tmp
for d in $1 $2; do
  (cd $R/$d && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab_$(basename $d)_prof -o x -- python3 bench.py --no-cpu-baseline > /dev/null 2>&1; echo "== $d"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/ab_$(basename $d)_prof/x_results.db 2>&1 | head -8 | cut -c1-100)
done
