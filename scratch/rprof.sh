#!/bin/bash
# kernel stats of a rollout-only run: rprof.sh <tag> [env assignments for the script]
TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/rprof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rprof_$TAG -- python3 $GRAFT_REPO_ROOT/scratch/rollout_prof.py 8192 40 20 5 > $GRAFT_REPO_ROOT/gpurun_out/rp_$TAG.txt 2>&1
cp $(find /tmp/rprof_$TAG -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/rollout_${TAG}_kernel_stats.csv
grep "^B=" $GRAFT_REPO_ROOT/gpurun_out/rp_$TAG.txt
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/rollout_${TAG}_kernel_stats.csv")))[:13]:
    print(f"{r['Name'].split('(')[0].replace('ltompc::','').replace('void ','')[:40]:42s} n={r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:8.1f} ms avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}")
PY
