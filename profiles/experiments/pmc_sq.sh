# SQ instruction / busy counters of the bench step's kernels (one --pmc pass; csv summarised per kernel)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --eager > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open("gpurun_out/pmc_sq/p_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("ags_k"): continue
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"]
print("| kernel | " + " | ".join(names) + " |"); print("|---|" + "---:|" * len(names))
for k, d in acc.items():
    print("| `%s` | " % k + " | ".join("%.3g" % (d[n][0] / max(d[n][1], 1)) if n in d else "-" for n in names) + " |")
PY
