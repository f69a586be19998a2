#!/bin/bash
# k_tail (scratch/tail_kernel.patch) for the DEEP tail only: launches of at most w instances, n iterations per launch
mkdir -p gpurun_out
for cfg in "0 8 4" "8 4 4" "8 8 4" "16 4 4" "16 8 4" "32 8 4" "16 8 1" "0 8 1"; do
  set -- $cfg
  LTOMPC_TAIL=$1 LTOMPC_TAIL_ITERS=$2 timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-profile --parts $3 > gpurun_out/tail.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/tail.json')); print('tail width $1 iters $2 parts $3:', round(d['value']), round(d['ms_per_step'],2), d['status_histogram_last_tick'])"
done
