#!/bin/bash
# round 5: a 200-step run of the headline (does the step time drift with temperature / clocks over 30 s?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5p
mkdir -p $O
timeout 600 python bench.py --steps 200 --warmup 10 --cpu-baseline none --no-modes --no-kernel-breakdown > $O/bench200.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench200.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['parity']['eps_max_abs_err'], d['roofline']['frac'], d['roofline']['clocks'])" | tee $O/summary.txt
