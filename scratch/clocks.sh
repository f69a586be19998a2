#!/bin/bash
# shader clock / power while a long rollout runs (sustained full-width load) vs the lock-step loop
cd $GRAFT_REPO_ROOT
(for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done) > gpurun_out/clocks.txt &
MON=$!
python3 scratch/rollout_prof.py 8192 40 100 5 > gpurun_out/clocks_run.txt 2>&1
kill $MON 2>/dev/null
grep "^B=" gpurun_out/clocks_run.txt
cat gpurun_out/clocks.txt | cut -c1-200 | head -45
