#!/bin/bash
# FC leg at several (samples, HHE_FC_CHUNK) pairs: tools/ab_fc2.sh "samples:chunk" ...
for sc in "$@"; do
  s=${sc%%:*}; ch=${sc##*:}
  echo -n "samples $s chunk $ch: "; HHE_FC_CHUNK=$ch timeout -k 10 300 python tools/fc_only.py $s 2>/dev/null | tail -1
done
