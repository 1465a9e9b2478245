#!/bin/bash
# round 5: smoke() and the driver's bench arguments on the shipped library (CPU leg off: measured in r5y, 424 s)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline none > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['parity']['eps_max_abs_err'], d['modes']['fast']['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['clocks'])" | tee $O/summary.txt
