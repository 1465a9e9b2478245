#!/bin/bash
# round 6: the stencil conv launch of the counter audit (tools/runs/r6j.sh) once more on the shipped library ad7d11af7522
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6q
mkdir -p $O
export TMPDIR=/tmp
for mode in conv ff1; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_WAVES SQ_LDS_DATA_FIFO_FULL" ; do
    tag=$(echo $set | cut -c1-20 | tr ' ' '_')
    bash tools/exp/pmc.sh final_${mode}_$tag $set -- python $GRAFT_REPO_ROOT/tools/exp/gemm_pmc.py $mode > $O/gemm_pmc_${mode}_$tag.txt 2>&1
  done
done
grep -h "pnc_gemm" $O/gemm_pmc_*.txt | grep -v splitk | grep "BUSY_CYCLES\|INSTS_MFMA\|INSTS_SALU\|INSTS_VALU\|INSTS_LDS\|BANK_CONFLICT\|IDX_ACTIVE\|SQ_WAVES" | cut -c1-30,70-150
