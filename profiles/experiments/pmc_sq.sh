# SQ instruction / busy counters of the bench step's kernels (one --pmc pass; csv summarised per kernel)
# usage: bash profiles/experiments/pmc_sq.sh <session tag>  -> gpurun_out/<tag>_sq_counters.md, gpurun_out/sq_counters.json
TAG=${1:-r00}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_sq
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --eager > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - $TAG <<'PY' | tee gpurun_out/${TAG}_sq_counters.md
import csv, collections, json, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open("gpurun_out/pmc_sq/p_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("ags_k"): continue
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"]
print(f"# {tag}: SQ counters per launch of the bench step's kernels (rocprofv3 --pmc, bench.py --eager; profiles/experiments/pmc_sq.sh)\n")
print("| kernel | " + " | ".join(names) + " |"); print("|---|" + "---:|" * len(names))
STAGE = [("ags_k_preprocess_bwd", "preprocess_bwd"), ("ags_k_preprocess", "preprocess"), ("ags_k_scan_tiles", "binning"),
         ("ags_k_bucket", "binning"), ("ags_k_tile_sort", "binning"), ("ags_k_render_fwd", "render_fwd"),
         ("ags_k_render_bwd", "render_bwd")]
stages = collections.defaultdict(lambda: collections.defaultdict(float))
def stage_of(k):
    for pat, st in STAGE:
        if k.startswith(pat): return st
# per STEP: a kernel counts by its launches per step (launches / the stage's most-launched kernel), so the one-off
# launches outside the steps (the workspace probe's general tile sort) are not a second kernel of the stage
steps = collections.defaultdict(int)
for k, d in acc.items():
    if stage_of(k): steps[stage_of(k)] = max(steps[stage_of(k)], max(v[1] for v in d.values()))
for k, d in acc.items():
    print("| `%s` | " % k + " | ".join("%.3g" % (d[n][0] / max(d[n][1], 1)) if n in d else "-" for n in names) + " |")
    st = stage_of(k)
    if st:
        for n in names:
            if n in d: stages[st][n] += d[n][0] / steps[st]
out = {k: dict(v) for k, v in stages.items()}
out["_session"] = tag
out["_note"] = "per step, summed over the kernels of a bench stage weighted by their launches per step; SQ_ACTIVE_INST_VALU is in quad-cycles summed over the SIMDs"
json.dump(out, open("gpurun_out/sq_counters.json", "w"), indent=1)
PY
rm -rf gpurun_out/pmc_sq
