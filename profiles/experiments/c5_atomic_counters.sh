#!/bin/bash
# Config 5's per-Gaussian forward kernel and tile sort: how many atomics with / without return reach the L2, how many go on
# to the fabric, and how the waves' time splits (the "are the returned slot atomics the wait?" question of VERDICT r05 #2).
# usage: bash profiles/experiments/c5_atomic_counters.sh <tag> -> gpurun_out/<tag>_c5_atomic_counters.md
TAG=${1:-r00}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
S=$R/profiles/experiments/c5_eager_steps.py
i=0
for set in "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_REQ_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/c5a_$i
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/c5a_$i -o p -- python3 $S > /dev/null 2> $R/gpurun_out/c5a_$i.log
done
cd $R
python3 - $TAG <<'PY' > gpurun_out/${TAG}_c5_atomic_counters.md
import csv, collections, sys, os, glob
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
names = []
for p in sorted(glob.glob("gpurun_out/c5a_*/p_counter_collection.csv")):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("ags_k"): continue
        if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(f"# {tag}: L2 / fabric request counters per launch, config 5 (5 M surfels @2048x2048, one view), eager steps "
      "(profiles/experiments/c5_atomic_counters.sh)\n")
print("| kernel | " + " | ".join(n.replace("_sum", "") for n in names) + " |"); print("|---|" + "---:|" * len(names))
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", [0, 1])[0]):
    print("| `%s` | " % k + " | ".join("%.4g" % (d[n][0] / max(d[n][1], 1)) if n in d else "-" for n in names) + " |")
PY
for j in 1 2 3 4 5; do tail -2 gpurun_out/c5a_$j.log >> gpurun_out/${TAG}_c5_atomic_counters.err 2>/dev/null; done
rm -rf gpurun_out/c5a_*
cat gpurun_out/${TAG}_c5_atomic_counters.md
