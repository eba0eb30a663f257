#!/bin/bash
# Two quick PMC passes (SQ issue / wait counters, LDS counters) over one bench step: tools/pmc_quick.sh <outdir under gpurun_out> [bench args]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o out --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --cpu-baseline 0 --extras 0 --kernel-timing 0 "$@" > $out/p$i.json 2> $out/p$i.log || echo "pass $i failed"
done
python3 $root/tools/pmc_summary.py $out 'ks_row|ntt_pass' > $out/summary.txt 2>&1
echo done
