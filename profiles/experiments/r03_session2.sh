cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python profiles/experiments/prof_dropin.py 680 1200 1 2>&1 | grep -v "^$" | head -60 > gpurun_out/r03_prof_dropin_c2.txt
python profiles/experiments/prof_dropin.py 512 512 8 2>&1 | grep -v "^$" | head -60 > gpurun_out/r03_prof_dropin_512.txt
head -4 gpurun_out/r03_prof_dropin_c2.txt; head -45 gpurun_out/r03_prof_dropin_512.txt | cut -c1-160
