#!/bin/bash
# round-3 final: the default bench line of the shipped library incl. the CPU oracle leg
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n
mkdir -p $O
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -n 3 $O/bench_default.err
python -c "import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['parity']['eps_max_abs_err'],d['modes']['fast']['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['cpu_baseline']['step_seconds'])"
