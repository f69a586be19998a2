#!/bin/bash
# kernel trace of a rollout-only run -> per-pass timeline summary: rtrace.sh <tag> [env assignments]
TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/rtrace_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/rtrace_$TAG -- python3 $GRAFT_REPO_ROOT/scratch/rollout_prof.py 8192 40 20 5 > $GRAFT_REPO_ROOT/gpurun_out/rt_$TAG.txt 2>&1
F=$(find /tmp/rtrace_$TAG -name "*kernel_trace.csv" | head -1)
grep "^B=" $GRAFT_REPO_ROOT/gpurun_out/rt_$TAG.txt
cp $F $GRAFT_REPO_ROOT/gpurun_out/rt_${TAG}_trace.csv; python3 $GRAFT_REPO_ROOT/scratch/${SUMMARY:-rtrace_summary.py} $F
