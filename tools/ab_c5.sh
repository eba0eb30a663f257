#!/bin/bash
# A/B of libraries on the config-5 rotation chain: tools/ab_c5.sh <rounds> lib1.so lib2.so ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    echo "$lib $(HHE_LIB=$lib timeout -k 10 200 python tools/config5_only.py 128 32 2>/dev/null | tail -1)"
  done
done
