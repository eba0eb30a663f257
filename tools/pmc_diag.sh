#!/bin/bash
# Diagnostic PMC passes (vector-memory pipeline and latency counters) over one bench step: tools/pmc_diag.sh <outdir> [bench args]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY" \
           "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $out/d$i -o out --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --cpu-baseline 0 --extras 0 --kernel-timing 0 "$@" > $out/d$i.json 2> $out/d$i.log || echo "pass $i failed"
  echo "pass $i finished"
done
echo done
