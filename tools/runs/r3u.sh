#!/bin/bash
# last check of the round: the driver's own sequence on the shipped tree (pytest -m gpu -x, smoke, default bench without the CPU leg),
# then two side timings (hipGraph replay of the step; CFG halves as independent stream pairs on the unfused path)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u
mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 400 python bench.py --cpu-baseline none > $O/bench_default_no_cpu.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench_default_no_cpu.json').read().strip().splitlines()[-1]);print('default', d['value'],d['ms_per_step'],d['parity']['eps_max_abs_err'],d['modes']['fast']['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))"
B="--steps 8 --warmup 2 --cpu-baseline none --no-modes --no-kernel-breakdown --no-parity"
for v in "eager:" "graph:--graph" "unfused:--no-fused-step" "unfused_split:--no-fused-step --split-samples" "eager2:"; do
  n=${v%%:*}; a=${v#*:}
  timeout 300 python bench.py $B $a 2>/dev/null | tail -1 > $O/bench_$n.json
  python -c "import json; d=json.loads(open('$O/bench_$n.json').read()); print('$n', round(d['ms_per_step'],2))"
done
