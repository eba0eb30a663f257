#!/bin/bash
# A/B of one environment knob under PMC counters on the NTT microbenchmark: tools/pmc_ab.sh KNOB "v1 v2" [counters...]
# (counters in their own rocprofv3 run with --kernel-trace only; output under gpurun_out/)
knob=$1; vals=$2; shift 2
ctrs=${@:-SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in $vals; do
  export $knob=$v
  rocprofv3 --pmc $ctrs --kernel-trace -d $root/gpurun_out/pmc_${knob}_$v -o out --output-format csv -- python3 $root/tools/ntt_micro.py 768 2 > $root/gpurun_out/pmc_${knob}_$v.log 2>&1 || exit 1
done
