#!/bin/bash
# A/B of environment knobs on the three legs (bench step, config-5 chain, FC), alternating on one box: tools/ab_all.sh <rounds> "VAR=1" "VAR=2" ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for kv in "$@"; do
    echo -n "$kv bench: "; env $kv timeout -k 10 200 python bench.py --cpu-baseline 0 --extras 0 --steps 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), round(d['roofline']['avg_launch_us'],1))"
    echo -n "$kv c5: "; env $kv timeout -k 10 200 python tools/config5_only.py 128 32 2>/dev/null | tail -1
    echo -n "$kv fc: "; env $kv timeout -k 10 200 python tools/fc_only.py 16 2>/dev/null | tail -1
  done
done
