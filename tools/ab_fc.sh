#!/bin/bash
# A/B of environment knobs on the FC leg alone, alternating runs on one box: tools/ab_fc.sh <rounds> "VAR=1" "VAR=2" ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for kv in "$@"; do
    echo -n "$kv: "; env $kv timeout -k 10 200 python tools/fc_only.py 16 2>/dev/null | tail -1
  done
done
