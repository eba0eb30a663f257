#!/bin/bash
# A/B of library builds on one box, alternating runs: tools/ab.sh <rounds> lib1.so lib2.so ...   (prints transcipherings/s and the row kernel's average)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    HHE_LIB=$lib timeout -k 10 200 python bench.py --cpu-baseline 0 --extras 0 --steps 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$lib', round(d['value'],1), round(d['roofline']['avg_launch_us'],1))"
  done
done
