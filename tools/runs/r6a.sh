#!/bin/bash
# round 6, first call: same-box baseline of the round-5 library (short bench), the CU-mask overlap probe, the SQ counter audit of
# the level-0 attention launches, the counter list of this box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_list.txt 2>&1)
timeout 300 python tools/exp/cu_mask_probe.py > $O/cu_mask_probe.log 2>&1
tail -12 $O/cu_mask_probe.log
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-baseline none --no-modes > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('bench', d['ms_per_step'], d['parity']['eps_max_abs_err'], d['roofline']['frac'], d['roofline'].get('clocks'))"
for mode in intra cross text; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SMEM" ; do
    tag=$(echo $set | cut -c1-20 | tr ' ' '_')
    bash tools/exp/pmc.sh attn_${mode}_$tag $set -- python $GRAFT_REPO_ROOT/tools/exp/attn_pmc.py $mode > $O/attn_pmc_${mode}_$tag.txt 2>&1
  done
done
grep -h "attn_views" $O/attn_pmc_*.txt | head -80
