#!/bin/bash
# Round 5: what do the blend backward's gradient-record atomics cost?
#  1. the counters the L2 / fabric side offers for atomics and writes (rocprofv3 -L), one pass per counter, config 5
#  2. same-box A/B against an experiment build whose atomic instruction is never executed (scratch/libags_noatom.so:
#     the condition is data-dependent and never true - everything else of the flush still runs), per kernel under
#     rocprofv3: config 5 (frozen scene), C2, the mapper loop
# The two builds (scratch/ is not tracked; made in the build container before the call):
#   scratch/libags_base5.so  = a copy of active-gs_amd/lib/libags_raster.so
#   scratch/libags_noatom.so = the same sources with, in render.hip's ags_k_render_bwd_mfma flush,
#       if (slot < nb && fld < 15) unsafeAtomicAdd(rec, ...)   ->   if (slot < nb && fld < 15 && (fld < 6 ? outg : outw) == 1.2345e38f) unsafeAtomicAdd(rec, ...)
#     compiled with the flags of active-gs_amd/build.py (copy csrc/ to scratch/exp_noatom, patch, hipcc each .hip, link)
# bash profiles/experiments/c5_atomics.sh <tag>   -> gpurun_out/<tag>_c5_atomics.md
TAG=${1:-r00}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in base5 noatom; do
  if [ ! -f $R/scratch/libags_$lib.so ]; then echo "c5_atomics.sh: scratch/libags_$lib.so is missing (see the header: both builds are made before the call)" >&2; exit 2; fi
done
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_c5_atomics.md
S=$R/profiles/experiments/c5_eager_steps.py
rocprofv3 -L > $R/gpurun_out/${TAG}_counters_list.txt 2>&1
echo "# $TAG: atomics of the blend backward (profiles/experiments/c5_atomics.sh)" > $OUT
echo >> $OUT; echo "## counters offered (rocprofv3 -L, names matching ATOMIC / WRREQ / EA0 / WRITE)" >> $OUT; echo '```' >> $OUT
grep -o -i -E "\b(TCC|TCP|TA|TD|SQ)_[A-Z0-9_]*(ATOMIC|WRREQ|WRITE|WR_)[A-Za-z0-9_]*" $R/gpurun_out/${TAG}_counters_list.txt | sort -u | tr '\n' ' ' >> $OUT
echo >> $OUT; echo '```' >> $OUT
echo >> $OUT; echo "## per-launch counters of \`ags_k_render_bwd_mfma<false>\`, config 5 (one pass per counter; a counter this build does not know is skipped)" >> $OUT
echo "| counter | per launch |" >> $OUT; echo "|---|---:|" >> $OUT
for c in TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_WRITE_sum TCC_WRITEBACK_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT WRITE_SIZE; do
  rm -rf $R/gpurun_out/c5a
  AGS_FREEZE=1 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/c5a -o p -- python3 $S > /dev/null 2>&1
  f=$R/gpurun_out/c5a/p_counter_collection.csv
  if [ -f $f ]; then
    python3 - $f $c >> $OUT <<'PY'
import csv, sys
tot = n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and r["Kernel_Name"].startswith("void ags_k_render_bwd_mfma"):
        tot += float(r["Counter_Value"]); n += 1
if n: print(f"| `{sys.argv[2]}` | {tot / n:.6g} |")
PY
  fi
done
rm -rf $R/gpurun_out/c5a
echo >> $OUT; echo "## A/B: product build vs the build whose atomic never executes (rocprofv3 kernel averages, us; interleaved)" >> $OUT
echo '```' >> $OUT
for rep in 1 2; do for tag in base5 noatom; do
  export AGS_LIB_PATH=$R/scratch/libags_$tag.so
  rm -rf $R/gpurun_out/c5a; AGS_FREEZE=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c5a -o x -- python3 $S > /dev/null 2>&1
  echo "config 5      $tag: $(python3 $R/profiles/rocpd_stats.py $R/gpurun_out/c5a/x_results.db 2>&1 | grep render_bwd | cut -c1-90)" >> $OUT
  rm -rf $R/gpurun_out/c5a; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c5a -o x -- python3 $R/bench.py --steps 300 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "C2            $tag: $(python3 $R/profiles/rocpd_stats.py $R/gpurun_out/c5a/x_results.db 2>&1 | grep render_bwd | cut -c1-90)" >> $OUT
  rm -rf $R/gpurun_out/c5a; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c5a -o x -- python3 $R/examples/mapper_loop.py > /dev/null 2>&1
  echo "mapper loop   $tag: $(python3 $R/profiles/rocpd_stats.py $R/gpurun_out/c5a/x_results.db 2>&1 | grep render_bwd | cut -c1-90)" >> $OUT
done; done
echo '```' >> $OUT
rm -rf $R/gpurun_out/c5a
cat $OUT
