# per-kernel A/B of library builds (scratch/libags_<tag>.so) on the FROZEN mapper-shaped workload (mapper_frozen_steps.py)
# bash profiles/experiments/ab_frozen_mapper.sh "<grep pattern>" tagA tagB ...
# PREREQUISITE: every tag needs its build scratch/libags_<tag>.so (scratch/ is not tracked: profiles/experiments/build_exp.py makes them)
pat=$1; shift
export R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  if [ ! -f $R/scratch/libags_$tag.so ]; then echo "ab_frozen_mapper.sh: scratch/libags_$tag.so is missing" >&2; exit 2; fi
done
python3 $R/profiles/experiments/mapper_frozen_steps.py make /tmp/frozen_map.pt 2>&1 | tail -1
for rep in 1 2; do for tag in "$@"; do
  export AGS_LIB_PATH=$R/scratch/libags_$tag.so
  rm -rf $R/gpurun_out/abf; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abf -o x -- python3 $R/profiles/experiments/mapper_frozen_steps.py run /tmp/frozen_map.pt > /tmp/abf.log 2>&1
  echo "== frozen mapper batch, $tag: $(tail -1 /tmp/abf.log)"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abf/x_results.db 2>&1 | grep -E "$pat" | cut -c1-100
  rm -rf $R/gpurun_out/abf
done; done
