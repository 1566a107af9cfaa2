#!/bin/bash
# Exact memory-side bytes per kernel (request-size counters, two passes) of a workload.
# usage: bash profiles/experiments/pmc_exact.sh <tag> <name> <out.json|-> -- <program> [args...]   (program: no wrappers, see gpurun notes)
TAG=$1; NAME=$2; OUT=$3; shift 4
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/pmcx_$NAME; rm -rf $D
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $D/rd -o p -- "$@" > $D.rd.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum --output-format csv -d $D/wr -o p -- "$@" > $D.wr.log 2>&1
cd $R
if [ "$OUT" = "-" ]; then python3 profiles/pmc_exact_summary.py $D --all-kernels > gpurun_out/${TAG}_${NAME}_pmc_exact.md
else python3 profiles/pmc_exact_summary.py $D $OUT ${TAG}_${NAME} > gpurun_out/${TAG}_${NAME}_pmc_exact.md; fi
tail -3 $D.rd.log >> gpurun_out/${TAG}_${NAME}_pmc_exact.err; rm -rf $D
cat gpurun_out/${TAG}_${NAME}_pmc_exact.md
