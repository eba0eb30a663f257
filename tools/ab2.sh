#!/bin/bash
# A/B of (library, environment) pairs on one box, alternating: tools/ab2.sh <rounds> "lib.so[,VAR=val]" ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    lib=${spec%%,*}; kv=${spec#*,}; [ "$kv" = "$spec" ] && kv="HHE_DUMMY=0"
    env HHE_LIB=$lib $kv timeout -k 10 200 python bench.py --cpu-baseline 0 --extras 0 --steps 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$spec', round(d['value'],1), round(d['roofline']['avg_launch_us'],1))"
  done
done
