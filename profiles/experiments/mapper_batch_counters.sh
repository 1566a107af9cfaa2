#!/bin/bash
# Counters of configuration 3's batched training iteration (11 views @512x512 of a mapper-grown map, learning rates 0:
# profiles/experiments/mapper_frozen_steps.py) - kernel durations, HBM traffic (separate passes) and two SQ passes.
# bash profiles/experiments/mapper_batch_counters.sh <tag>  -> gpurun_out/<tag>_mapper_batch_counters.md
TAG=${1:-r00}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
S=$R/profiles/experiments/mapper_frozen_steps.py
OUT=$R/gpurun_out/${TAG}_mapper_batch_counters.md
python3 $S make /tmp/frozen_map.pt 2>&1 | tail -1 > /tmp/frozen_make.txt
echo "# $TAG: configuration 3's batched iteration (11 views @512x512) on a frozen mapper-grown map: $(cat /tmp/frozen_make.txt)" > $OUT
echo >> $OUT; echo "## kernel durations (rocprofv3 --kernel-trace --stats, 120 iterations)" >> $OUT
rm -rf $R/gpurun_out/mbc; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/mbc -o k -- python3 $S run /tmp/frozen_map.pt > /dev/null 2>&1
python3 $R/profiles/rocpd_stats.py $R/gpurun_out/mbc/k_results.db 2>&1 | head -12 | cut -c1-120 >> $OUT
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/mbc_$tag
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/mbc_$tag -o p -- python3 $S run /tmp/frozen_map.pt 40 > /dev/null 2>&1
done
python3 - $R >> $OUT <<'PY'
import csv, collections, sys, os, glob
R = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
names = []
for p in sorted(glob.glob(f"{R}/gpurun_out/mbc_*/p_counter_collection.csv")):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("ags_k"): continue
        if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("\n## counters per launch (FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE counts wide reads at half their bytes on gfx950)\n")
print("| kernel | " + " | ".join(names) + " |"); print("|---|" + "---:|" * len(names))
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", [0, 1])[0]):
    print("| `%s` | " % k + " | ".join("%.4g" % (d[n][0] / max(d[n][1], 1)) if n in d else "-" for n in names) + " |")
PY
rm -rf $R/gpurun_out/mbc $R/gpurun_out/mbc_*
cat $OUT
