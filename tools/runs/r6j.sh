#!/bin/bash
# round 6: SQ counter audit of the long-K GEMM main loops (FF2 level 2 on gemm_glds_kernel, FF1 level 2 on the persistent GEGLU kernel,
# level-0 3x3 conv on the stencil-tile kernel): LDS port busy, bank conflicts, MFMA busy, waits
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j
mkdir -p $O
export TMPDIR=/tmp
for mode in ff2 ff1 conv; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_WAVES SQ_LDS_DATA_FIFO_FULL" ; do
    tag=$(echo $set | cut -c1-20 | tr ' ' '_')
    bash tools/exp/pmc.sh gemm_${mode}_$tag $set -- python $GRAFT_REPO_ROOT/tools/exp/gemm_pmc.py $mode > $O/gemm_pmc_${mode}_$tag.txt 2>&1
  done
done
grep -h "pnc_gemm" $O/gemm_pmc_*.txt | grep -v splitk | cut -c1-40,70-160
