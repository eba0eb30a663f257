#!/bin/bash
# Round measurement set on one MI355X, all from the same box: the rocprofv3 kernel stats of the bench command, the PMC passes
# (their summary goes straight into profiles/ so that the bench line below is stamped with the current source hash), then the
# bench line itself.  Writes under gpurun_out/final/ ; copy stats/out_kernel_stats.csv, pmc_summary.json and
# bench_default.json into profiles/ afterwards.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/final
mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $out/stats -o out --output-format csv -- python3 $root/bench.py --cpu-baseline 0 --extras 0 > $out/bench_under_rocprof.json 2> $out/stats.log ) || exit 1
bash $root/tools/pmc_passes.sh final/pmc --batch 128 > $out/pmc.log 2>&1 || exit 1
python3 $root/tools/pmc_collect.py $out/pmc 128 1 $out/pmc_summary.json > $out/pmc_collect.log 2>&1 || exit 1
cp $out/pmc_summary.json $root/profiles/r2_pmc_summary.json
timeout -k 10 500 python3 $root/bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
echo done
