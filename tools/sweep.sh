#!/bin/bash
# A/B sweep of the internal-stream / chunk knobs in ONE process-per-variant on the same device
cd "$(dirname "$0")/.."
for cfg in "0 256" "1 32" "1 64" "2 16" "2 32" "2 64" "3 32" "4 16" "4 32"; do
  set -- $cfg
  HHE_STREAMS=$1 HHE_CHUNK=$2 timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-baseline 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('streams=$1 chunk=$2', round(d['value'],1), 'tr/s', round(d['ms_per_step'],1),'ms')"
done
