#!/bin/bash
# Round-end measurement set on one MI355X: bench line, rocprofv3 kernel stats of the same command, PMC traffic passes.
# Writes under gpurun_out/final/ ; copy the summaries into profiles/ afterwards.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/final
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $root/bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
rocprofv3 --kernel-trace --stats -d $out/stats -o out --output-format csv -- python3 $root/bench.py --cpu-baseline 0 > $out/bench_under_rocprof.json 2> $out/stats.log || exit 1
# counters in their own passes (bench.py runs the library default: eager launching, no graph replay)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o out --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --cpu-baseline 0 > $out/pmc_fetch.json 2> $out/pmc_fetch.log || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o out --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --cpu-baseline 0 > $out/pmc_write.json 2> $out/pmc_write.log || exit 1
echo done
