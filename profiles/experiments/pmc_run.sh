# HBM traffic counters of the bench step, one counter per pass (MI355X_MICROARCH.md: separate --pmc runs)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --eager > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 profiles/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/p_counter_collection.csv gpurun_out/pmc_hbm_bytes.json
