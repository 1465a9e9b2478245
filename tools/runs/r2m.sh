#!/bin/bash
# PMC passes over the level-0 attention kernel: VALU-busy / MFMA-busy / wait cycles (run in round 2 with an experimental ping-pong
# schedule as second arm: argument 1; the shipped kernel ignores it)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_TRANS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  for pp in 0 1; do
    (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/apmc_${i}_$pp -- python $GRAFT_REPO_ROOT/tools/exp/attn_pmc.py $pp > $GRAFT_REPO_ROOT/gpurun_out/r2m/log_${i}_$pp.txt 2>&1)
    f=$(find /tmp/apmc_${i}_$pp -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" $pp <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    if "attn_views" in r["Kernel_Name"]:
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(agg.items()):
    print(f"pingpong={sys.argv[2]} {k:28s} {v / max(n, 1):16.0f} per launch ({n} launches)")
P
  done
  i=$((i+1))
done | tee gpurun_out/r2m/attn_pmc_summary.txt
