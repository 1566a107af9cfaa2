# HBM traffic counters of the bench step, one counter per pass (MI355X_MICROARCH.md: separate --pmc runs)
# usage: bash profiles/experiments/pmc_run.sh <session tag>   -> gpurun_out/pmc_hbm_bytes.json (copy into profiles/)
TAG=${1:-r00}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --eager > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 profiles/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/p_counter_collection.csv gpurun_out/pmc_hbm_bytes.json $TAG | tee gpurun_out/${TAG}_pmc_hbm.md
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
