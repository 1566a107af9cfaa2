# Round 5: the 16-lane-group blend loops (AgsTuning.blend_group = 16: four surfels in flight per wave) against the default
# (one surfel per wave and iteration), per kernel under rocprofv3, interleaved repetitions on one box:
#   C2 bench step, the mapper loop @512x512 (configuration 3: the reference's own shape), configuration 4's share, configuration 5
# bash profiles/experiments/ab_blend_group.sh [reps]      -> stdout (tee into gpurun_out/)
# PREREQUISITE: the tree has no blend_group switch any more - the 16-lane-group loops were measured, lost everywhere and
# were taken out; they survive as profiles/experiments/r05_blend_group16.patch.  Apply that patch and
# rebuild before running this script, or it compares two identical builds.
reps=${1:-2}
export R=$GRAFT_REPO_ROOT
if ! grep -q "blend_group" $R/include/ags_raster.h; then
  echo "ab_blend_group.sh: include/ags_raster.h has no blend_group field - apply profiles/experiments/r05_blend_group16.patch and rebuild first" >&2; exit 2
fi; cd /tmp && export TMPDIR=/tmp
pat="render_fwd|render_bwd"
run() { # label, command...
  label=$1; shift
  rm -rf $R/gpurun_out/abg; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abg -o x -- "$@" > /dev/null 2>&1
  echo "== $label, AGS_BLEND_GROUP=$AGS_BLEND_GROUP"; python3 $R/profiles/rocpd_stats.py $R/gpurun_out/abg/x_results.db 2>&1 | grep -E "$pat" | cut -c1-110
  rm -rf $R/gpurun_out/abg
}
for rep in $(seq $reps); do for val in 64 16; do
  export AGS_BLEND_GROUP=$val
  run "C2 bench" python3 $R/bench.py --steps 300 --no-cpu-baseline --no-extras
  run "mapper loop 512x512" python3 $R/examples/mapper_loop.py
  AGS_FREEZE=1 run "config 4 share" python3 $R/examples/large_configs.py --only c4 --steps 10
  AGS_FREEZE=1 run "config 5" python3 $R/profiles/experiments/c5_eager_steps.py
done; done
# end-to-end: the mapper loop's wall time, un-profiled, three runs each
for val in 64 16 64 16 64 16; do
  echo "== mapper loop seconds, AGS_BLEND_GROUP=$val: $(AGS_BLEND_GROUP=$val python3 $R/examples/mapper_loop.py 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["seconds"], d["final_surfels"])')"
done
