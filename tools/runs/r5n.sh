#!/bin/bash
# round 5: the driver's exact default command on the final tree (with the CPU leg)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n
mkdir -p $O
t0=$(date +%s)
timeout 580 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $? $(( $(date +%s) - t0 )) s" | tee $O/times.txt
python -c "import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print('default', d['value'],d['ms_per_step'],d['parity']['eps_max_abs_err'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline']['clocks']['sclk_mhz_median'], d['cpu_baseline']['value'], d['cpu_baseline']['step_seconds'])" | tee -a $O/times.txt
