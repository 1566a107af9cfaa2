# second SQ counter pass: where the waves wait (LDS, memory, issue) - bash profiles/experiments/pmc_sq2.sh <tag>
TAG=${1:-r00}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; do
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2
rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --eager > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2.log 2>&1
python3 - $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2/p_counter_collection.csv <<'PY'
import csv, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("ags_k"): continue
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k[:34].ljust(34), {n.replace("SQ_", ""): float("%.3g" % (v[0] / max(v[1], 1))) for n, v in d.items()})
PY
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2
