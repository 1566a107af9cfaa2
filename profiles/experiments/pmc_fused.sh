cd /tmp && export TMPDIR=/tmp AGS_BENCH_EAGER_PIPELINE=1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --eager > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 profiles/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/p_counter_collection.csv gpurun_out/pmc_fused.json r02_c_pipelined | tee gpurun_out/r02_c_pmc_hbm_pipelined.md
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
