#!/bin/bash
# PMC passes over kbench (stages given in $STAGES, default fwd,bwd)
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
ST=${STAGES:-fwd,bwd}
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  bash tools/gpu_pmc.sh "$C" --side ${SIDE:-64} --reps 2 --stages $ST --mask on 2>&1 | grep -v "amdgpu.ids" | tee -a gpurun_out/pmc_r2_$i.txt
done
