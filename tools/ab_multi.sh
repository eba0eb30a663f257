#!/bin/bash
# A/B of libraries on three workloads, alternating: config-5 rotation chain, FC 784x10, config-2 bench.  tools/ab_multi.sh <rounds> lib1.so lib2.so ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    echo "$lib | $(HHE_LIB=$lib timeout -k 10 200 python tools/config5_only.py 128 32 2>/dev/null | tail -1 | sed 's/.*batch 32: //')"
    echo "$lib | $(HHE_LIB=$lib timeout -k 10 200 python tools/fc_only.py 16 2>/dev/null | tail -1)"
    HHE_LIB=$lib timeout -k 10 200 python bench.py --cpu-baseline 0 --extras 0 --steps 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$lib | bench', round(d['value'],1), round(d['roofline']['avg_launch_us'],1))"
  done
done
