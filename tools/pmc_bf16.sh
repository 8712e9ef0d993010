#!/bin/bash
# SQ occupancy / stall counters of the bf16 conv kernels (three --pmc passes, no trace domains): the evidence behind DESIGN.md 3.1b
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bf16
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python3 $ROOT/bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events"
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -- $B > $OUT/$tag.log 2>&1
done
du -sh $OUT
