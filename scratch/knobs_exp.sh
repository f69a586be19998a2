#!/bin/bash
# scheduling-only knobs with four handles per GPU: switch-over width, poll interval, re-pack threshold
mkdir -p gpurun_out
for cfg in "128 4 6" "96 4 6" "192 4 6" "256 4 6" "128 3 6" "128 4 5" "128 4 7" "160 3 6" "128 3 7" "64 4 6" "128 4 6"; do
  set -- $cfg
  LTOMPC_PACK_NUM=$3 timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --narrow-width $1 --poll-every $2 > gpurun_out/kn.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/kn.json')); print('narrow $1 poll $2 pack $3/8:', round(d['value']), round(d['ms_per_step'],2))"
done
