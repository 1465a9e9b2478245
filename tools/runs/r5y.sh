#!/bin/bash
# round 5: the driver's own commands on the shipped library (default bench incl. the CPU leg, smoke), and the three stage benches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5y
mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $? $(( $(date +%s) - t0 )) s" | tee -a $O/times.txt
t0=$(date +%s)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $? $(( $(date +%s) - t0 )) s" | tee -a $O/times.txt
python -c "import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print('default', d['value'],d['ms_per_step'],d['parity']['eps_max_abs_err'], d['roofline']['frac'], d['roofline'].get('traffic'), d['cpu_baseline'])" | tee -a $O/times.txt
for st in vae-decode vae-encode text-tower; do
  t0=$(date +%s)
  timeout 300 python bench.py --stage $st > $O/bench_$st.json 2> $O/bench_$st.err; echo "$st rc $? $(( $(date +%s) - t0 )) s" | tee -a $O/times.txt
  tail -c 600 $O/bench_$st.json
done
t0=$(date +%s)
timeout 400 python bench.py --hoist --cpu-baseline none --no-modes > $O/bench_hoist.json 2> $O/bench_hoist.err; echo "hoist rc $? $(( $(date +%s) - t0 )) s" | tee -a $O/times.txt
python -c "import json;d=json.loads(open('$O/bench_hoist.json').read().strip().splitlines()[-1]);print('hoist', d['value'],d['ms_per_step'])" | tee -a $O/times.txt
