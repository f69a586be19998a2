#!/bin/bash
# Measurements behind profiles/r03/<tag>_*: bench JSON (the driver's shape), rocprofv3 kernel stats of the same command,
# HBM traffic (two PMC passes with the SAME command shape as the bench line, counters only: no trace domains beside them).  usage: profile_r03.sh <tag> [bench flags]
set -e
TAG=${1:-v1}; shift || true
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 "$@" > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > $O/${TAG}_stats.log 2>&1; echo "stats done"
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-extras "$@" > $O/${TAG}_pmc_fetch.log 2>&1; echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-extras "$@" > $O/${TAG}_pmc_write.log 2>&1; echo "write done"
python3 $R/profiles/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.txt
tail -25 $O/${TAG}_pmc_traffic.txt
head -12 $O/${TAG}_kernel_stats.csv
