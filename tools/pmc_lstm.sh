#!/bin/bash
# GPU box: SQ / TCP counters of the conv-LSTM kernel for both tile heights (per-layer launch path)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for m in 1111111 2222222; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
             "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
    tag=$(echo $set | cut -d' ' -f1)
    VF_LSTM_MREP=$m rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_${m}_$tag -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${m}_$tag.log 2>&1
  done
done
ls $R/gpurun_out | grep pmc | head -20
