#!/bin/bash
# round 5: PMC traffic passes again with the step / setup split of tools/pmc_traffic.py (same library)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5u
mkdir -p $O
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
BENCH1="python $GRAFT_REPO_ROOT/bench.py $ARGS"
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- $BENCH1 > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1)
done
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 2 precise "bench.py $ARGS" > $O/pmc.log 2>&1
python tools/pmc_traffic.py --mfma /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES 2 precise 170.0 >> $O/pmc.log 2>&1
mkdir -p $O/pmc && cp profiles/round5/pmc_traffic.json profiles/round5/pmc_precise_*_by_kernel.txt $O/pmc/
head -20 $O/pmc.log
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-baseline none --no-modes --no-kernel-breakdown > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_note'])"
