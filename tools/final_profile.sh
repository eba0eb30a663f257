#!/bin/bash
# Round measurement set on one MI355X, all from the same box: the rocprofv3 kernel stats of the bench command, the PMC passes
# (their summary goes straight into profiles/ so that the bench line below is stamped with the current source hash), the bench
# line itself, then the kernel stats of the config-5 chain and of the FC, the HIP API trace of 1 vs 4 key-set requests and the
# mnist-e2e workload.  Writes under gpurun_out/final/ ; tools/final_copy.sh copies the summaries into profiles/.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/final
mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $out/stats -o out --output-format csv -- python3 $root/bench.py --cpu-baseline 0 --extras 0 > $out/bench_under_rocprof.json 2> $out/stats.log ) || exit 1
bash $root/tools/pmc_passes.sh final/pmc --batch 128 > $out/pmc.log 2>&1 || exit 1
python3 $root/tools/pmc_collect.py $out/pmc 128 1 $out/pmc_summary.json > $out/pmc_collect.log 2>&1 || exit 1
cp $out/pmc_summary.json $root/profiles/r3_pmc_summary.json
timeout -k 10 800 python3 $root/bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
timeout -k 10 300 python3 $root/bench.py --workload mnist-e2e --steps 1 --warmup 1 > $out/bench_mnist_e2e.json 2> $out/bench_mnist_e2e.err || exit 1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $out/c5 -o out --output-format csv -- python3 $root/tools/config5_only.py 64 32 > $out/c5.log 2>&1 )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $out/fc -o out --output-format csv -- python3 $root/tools/fc_only.py 16 > $out/fc.log 2>&1 )
bash $root/tools/pmc_config5.sh final/c5pmc > $out/c5pmc.log 2>&1
for r in 1 4; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --hip-trace --memory-copy-trace --stats -d $out/ks$r -o out --output-format csv -- python3 $root/tools/keyset_requests.py $r > $out/ks$r.log 2>&1 )
done
echo done
