#!/bin/bash
# Round 5: why the per-Gaussian forward kernel sits at 0.13 of the HBM peak at C2 - where its wave-cycles go
# (two SQ passes + one TCP pass over eager bench steps), and the cull-first kernel forced at C2's size for comparison.
# bash profiles/experiments/pmc_preprocess_c2.sh <tag>  -> gpurun_out/<tag>_preprocess_c2.md
TAG=${1:-r00}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_preprocess_c2.md
echo "# $TAG: the per-Gaussian forward kernel at C2 (200 k surfels @1200x680), counters per launch (profiles/experiments/pmc_preprocess_c2.sh)" > $OUT
echo >> $OUT; echo "| counter | ags_k_preprocess<2> | ags_k_preprocess_bwd_rows<1> | ags_k_render_fwd<1> |" >> $OUT; echo "|---|---:|---:|---:|" >> $OUT
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf $R/gpurun_out/ppc
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/ppc -o p -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras --eager > /dev/null 2>&1
  f=$R/gpurun_out/ppc/p_counter_collection.csv
  [ -f $f ] && python3 - $f >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    a = acc[r["Counter_Name"]][k]; a[0] += float(r["Counter_Value"]); a[1] += 1
for c, d in acc.items():
    def v(prefix):
        for k, (t, n) in d.items():
            if k.startswith(prefix): return "%.4g" % (t / n)
        return "-"
    print(f"| `{c}` | {v('ags_k_preprocess<2')} | {v('ags_k_preprocess_bwd_rows')} | {v('ags_k_render_fwd<1')} |")
PY
done
rm -rf $R/gpurun_out/ppc
echo >> $OUT; echo "## the cull-first kernel forced at C2 (AGS_PRE_CULL_MIN_N=0) against the default, stage medians of bench.py (ms), interleaved" >> $OUT; echo '```' >> $OUT
cd $R
for rep in 1 2 3; do for v in default 0; do
  if [ $v = default ]; then unset AGS_PRE_CULL_MIN_N; else export AGS_PRE_CULL_MIN_N=$v; fi
  python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('AGS_PRE_CULL_MIN_N=$v', 'ms_per_step', round(d['ms_per_step'],5), 'preprocess', d['config']['stage_ms']['preprocess'], 'preprocess_bwd', d['config']['stage_ms']['preprocess_bwd'])" >> $OUT
done; done
echo '```' >> $OUT
cat $OUT
