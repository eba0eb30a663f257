#!/bin/bash
# A/B of environment knobs on one box, alternating runs: tools/ab_env.sh <rounds> "VAR=1" "VAR=2" ...  (product library)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for kv in "$@"; do
    env $kv timeout -k 10 200 python bench.py --cpu-baseline 0 --extras 0 --steps 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$kv', round(d['value'],1), round(d['roofline']['avg_launch_us'],1))"
  done
done
