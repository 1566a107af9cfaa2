# SQ counters of render_bwd variants on bench (C2) and the mapper loop
cd /tmp && export TMPDIR=/tmp R=$GRAFT_REPO_ROOT
summ() { python3 - "$1" <<'PY'
import csv, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("ags_k_render_bwd"): continue
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k, {n: float("%.3g" % (v[0] / max(v[1], 1))) for n, v in d.items()})
PY
}
for m in 0 2; do export AGS_BWD_MFMA=$m
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR"; do
rm -rf $R/gpurun_out/pmc_b; rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_b -o p -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --eager > /dev/null 2>&1
echo "C2 mfma=$m"; summ $R/gpurun_out/pmc_b/p_counter_collection.csv
rm -rf $R/gpurun_out/pmc_b; rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_b -o p -- python3 $R/examples/mapper_loop.py --keyframes 12 > /dev/null 2>&1
echo "mapper mfma=$m"; summ $R/gpurun_out/pmc_b/p_counter_collection.csv
done; done
rm -rf $R/gpurun_out/pmc_b
