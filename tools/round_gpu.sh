#!/bin/bash
# one GPU call of the round: full GPU tests, the default bench line, kernel-stat profiles of the config-5 chain and of the FC
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$1; mkdir -p $out
cd $root
python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log
timeout -k 10 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; python -c "
import json; d=json.load(open('$out/bench_default.json')); print('bench', d['value'], d['roofline'].get('avg_launch_us')); e=d['extras']
print('config5', e['config5_rotate_chain']['achieved_GBps'], 'mnist dec/fc ms', e['mnist_1fc']['decompose_ms_per_sample'], e['mnist_1fc']['fc_ms_per_sample'], 'refdef', e['reference_defaults']['transcipherings_per_s'], 'latency', e['per_record_latency'])"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $out/c5 -o out --output-format csv -- python3 $root/tools/config5_only.py 64 32 > $out/c5.log 2>&1 ); tail -1 $out/c5.log
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $out/fc -o out --output-format csv -- python3 $root/tools/fc_only.py 16 > $out/fc.log 2>&1 ); tail -1 $out/fc.log
find $out -name "*kernel_stats.csv" | head
