#!/bin/bash
# round-3 final measurement pass, part 2: PMC passes on timed work only (--no-parity: 1 warm-up + 1 step = 2 evaluations), then the
# default bench line incl. the CPU oracle leg
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j
mkdir -p $O
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
BENCH1="python $GRAFT_REPO_ROOT/bench.py $ARGS"
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- $BENCH1 > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1)
done
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 2 precise "bench.py $ARGS" > $O/pmc.log 2>&1
python tools/pmc_traffic.py --mfma /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES 2 precise 178.0 >> $O/pmc.log 2>&1
mkdir -p $O/pmc && cp profiles/round3/pmc_* $O/pmc/ 2>/dev/null
head -16 $O/pmc.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -n 4 $O/bench_default.err
python -c "import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['parity']['eps_max_abs_err'],d['modes']['fast']['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['cpu_baseline'])"
