#!/bin/bash
# Config 5 (5 M surfels @2048x2048, one view) under rocprofv3 with the CURRENT kernels: kernel durations, HBM traffic
# (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters (two passes) of eagerly launched steps.
# usage: bash profiles/experiments/c5_counters.sh <tag>  -> gpurun_out/<tag>_c5_{kernel_stats,pmc_hbm,sq_counters}.md
TAG=${1:-r00}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
S=$R/profiles/experiments/c5_eager_steps.py
rm -rf $R/gpurun_out/c5k; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c5k -o k -- python3 $S > $R/gpurun_out/${TAG}_c5_steps.log 2>&1
python3 $R/profiles/rocpd_stats.py $R/gpurun_out/c5k/k_results.db 2>&1 | head -10 | cut -c1-120 > $R/gpurun_out/${TAG}_c5_kernel_stats.md
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/c5_$c
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/c5_$c -o p -- python3 $S > /dev/null 2>&1
done
python3 $R/profiles/pmc_summary.py $R/gpurun_out/c5_FETCH_SIZE/p_counter_collection.csv $R/gpurun_out/c5_WRITE_SIZE/p_counter_collection.csv $R/gpurun_out/pmc5_hbm_bytes.json ${TAG}_c5 > $R/gpurun_out/${TAG}_c5_pmc_hbm.md
rm -rf $R/gpurun_out/c5_sq1 $R/gpurun_out/c5_sq2
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/c5_sq1 -o p -- python3 $S > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/c5_sq2 -o p -- python3 $S > $R/gpurun_out/${TAG}_c5_sq2.log 2>&1
cd $R
python3 - $TAG <<'PY' > gpurun_out/${TAG}_c5_sq_counters.md
import csv, collections, sys, os
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
names = []
for d in ("c5_sq1", "c5_sq2"):
    p = f"gpurun_out/{d}/p_counter_collection.csv"
    if not os.path.exists(p): continue
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("ags_k"): continue
        if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(f"# {tag}: SQ counters per launch, config 5 (5 M surfels @2048x2048, one view), eager steps (profiles/experiments/c5_counters.sh)\n")
print("| kernel | " + " | ".join(names) + " |"); print("|---|" + "---:|" * len(names))
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", [0, 1])[0]):
    print("| `%s` | " % k + " | ".join("%.4g" % (d[n][0] / max(d[n][1], 1)) if n in d else "-" for n in names) + " |")
PY
rm -rf gpurun_out/c5k gpurun_out/c5_FETCH_SIZE gpurun_out/c5_WRITE_SIZE gpurun_out/c5_sq1 gpurun_out/c5_sq2
head -12 gpurun_out/${TAG}_c5_kernel_stats.md; cat gpurun_out/${TAG}_c5_pmc_hbm.md; cat gpurun_out/${TAG}_c5_sq_counters.md; tail -3 gpurun_out/${TAG}_c5_steps.log
