#!/bin/bash
# narrow-kernel thresholds (LTOMPC_RIC1 / LTOMPC_STEP1) against the number of handles per GPU
mkdir -p gpurun_out
for cfg in "4 4 512 512" "4 4 256 256" "4 4 128 128" "4 4 64 64" "4 4 256 512" "4 4 128 512" "8 8 128 128" "8 8 64 64" "8 8 256 256" "6 8 128 128" "2 4 256 256" "1 4 256 256"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$2 LTOMPC_RIC1=$3 LTOMPC_STEP1=$4 timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --parts $1 > gpurun_out/thr.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/thr.json')); print('parts $1 queues $2 ric1 $3 step1 $4:', round(d['value']), round(d['ms_per_step'],2))"
done
