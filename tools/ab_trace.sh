#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of one short bench run per library: tools/ab_trace.sh lib1.so lib2.so ...
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for lib in "$@"; do
  tag=$(basename $lib .so)
  out=$root/gpurun_out/abtrace_$tag
  export HHE_LIB=$root/$lib
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $out -o out --output-format csv -- python3 $root/bench.py --cpu-baseline 0 --extras 0 --steps 2 > $out.json 2> $out.log )
  echo "== $lib $(python3 -c "import json;print(round(json.load(open('$out.json'))['value'],1))")"
  head -6 $out/out_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
