#!/bin/bash
# round 3, GPU session 1: the whole GPU suite with the parity log, the bench line, the drop-in path, hip-trace of the
# drop-in loop (synchronisations per call), counter-stride variants and HBM counters on config 5.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export AGS_PARITY_LOG=$R/gpurun_out/r03_parity_log.jsonl; rm -f $AGS_PARITY_LOG
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r03_pytest.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r03_pytest.log
unset AGS_PARITY_LOG
timeout 1200 python bench.py > gpurun_out/r03_a_bench.json 2> gpurun_out/r03_a_bench.err; echo "bench rc $?"; cut -c1-400 gpurun_out/r03_a_bench.json; tail -5 gpurun_out/r03_a_bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r03_a_bench_driver_shape.json 2>/dev/null; cut -c1-300 gpurun_out/r03_a_bench_driver_shape.json
timeout 600 python examples/dropin_path.py 2>&1 | tail -2 > gpurun_out/r03_a_dropin.json; cut -c1-400 gpurun_out/r03_a_dropin.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/hipt; timeout 600 rocprofv3 --hip-trace --stats -d $R/gpurun_out/hipt -o h -- python3 $R/profiles/experiments/prof_dropin.py > $R/gpurun_out/r03_hiptrace.log 2>&1
python3 - <<'PY' > $R/gpurun_out/r03_a_dropin_hip_api_stats.md 2>&1
import sqlite3, glob, os
db = glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/hipt/**/*_results.db", recursive=True)
con = sqlite3.connect(db[0])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
cand = [t for t in tabs if "top" in t.lower() or "hip" in t.lower()]
print("tables:", cand)
for t in cand:
    if "top" in t.lower() and "kernel" not in t.lower():
        try:
            cols = [c[1] for c in con.execute(f"pragma table_info({t})")]
            print("##", t, cols)
            for r in con.execute(f"select * from {t} limit 40"): print("|", " | ".join(str(x) for x in r), "|")
        except Exception as e: print(t, e)
PY
head -60 $R/gpurun_out/r03_a_dropin_hip_api_stats.md
cd $R
bash profiles/experiments/ab_kernel_large.sh "preprocess|tile_sort|render" cur tc8 tc32 2>&1 | tee gpurun_out/r03_tc_stride_c5.txt
for t in cur tc8 tc32; do if [ $t = cur ]; then unset AGS_LIB_PATH; else export AGS_LIB_PATH=$R/scratch/libags_$t.so; fi; python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['config']['stage_ms'])"; done 2>&1 | tee gpurun_out/r03_tc_stride_c2.txt
unset AGS_LIB_PATH
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc5_$c
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc5_$c -o p -- python3 $R/examples/large_configs.py --only c5 --steps 4 > $R/gpurun_out/pmc5_$c.log 2>&1
done
cd $R
python3 profiles/pmc_summary.py gpurun_out/pmc5_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc5_WRITE_SIZE/p_counter_collection.csv gpurun_out/pmc5_hbm_bytes.json r03_a_c5 | tee gpurun_out/r03_a_c5_pmc_hbm.md
rm -rf gpurun_out/pmc5_FETCH_SIZE gpurun_out/pmc5_WRITE_SIZE gpurun_out/hipt
