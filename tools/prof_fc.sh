#!/bin/bash
# kernel-stat profile of the FC leg alone: tools/prof_fc.sh <tag>  -> gpurun_out/<tag>/fc_kernel_stats.txt
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$1; mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $out/fc -o out --output-format csv -- python3 $root/tools/fc_only.py 16 > $out/fc.log 2>&1 ); tail -1 $out/fc.log
f=$(find $out/fc -name "*kernel_stats.csv" | head -1); python $root/tools/kstats.py $f > $out/fc_kernel_stats.txt; cp $f $out/fc_kernel_stats.csv; rm -rf $out/fc; cat $out/fc_kernel_stats.txt
