#!/bin/bash
# kernel-stat profile of one-record calls: tools/prof_one.sh <tag>
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$1; mkdir -p $out
python3 $root/tools/one_record.py 8 | tail -1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $out/one -o out --output-format csv -- python3 $root/tools/one_record.py 8 > $out/one.log 2>&1 ); tail -1 $out/one.log
f=$(find $out/one -name "*kernel_stats.csv" | head -1); python $root/tools/kstats.py $f > $out/one_kernel_stats.txt; rm -rf $out/one; cat $out/one_kernel_stats.txt
