#!/bin/bash
# round 5, call 13: do independent stream pairs per CFG half pay?  (--no-fused-step so that --split-samples is live), order A B B A
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l
mkdir -p $O
export TMPDIR=/tmp
B="--steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity --no-fused-step"
for opt in "" "--split-samples" "--split-samples" ""; do
    timeout 400 python bench.py $B $opt > $O/bench.json 2> $O/bench.err
    python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('[$opt]', round(d['ms_per_step'],2))" | tee -a $O/ab.log
done
