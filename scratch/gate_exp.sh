#!/bin/bash
# wide-phase gate experiment: LTOMPC_GATE=capacity LTOMPC_GATE_W=release width, parts P
mkdir -p gpurun_out
for cfg in "2 0 512" "2 1 512" "3 1 512" "4 1 512" "4 2 512" "3 2 512" "2 1 2048" "3 1 2048" "4 2 2048" "4 1 128"; do
  set -- $cfg
  LTOMPC_GATE=$2 LTOMPC_GATE_W=$3 timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --parts $1 > gpurun_out/gate.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/gate.json')); print('parts $1 gate $2 w $3:', round(d['value']), round(d['ms_per_step'],2))"
done
