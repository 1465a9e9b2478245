#!/bin/bash
# round 5, call 9: reworked concat-with-records kernel (channel slices, adaptive chunk): test + micro-benchmark + whole-step bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest -q --timeout=500 tests/test_kernels_gpu.py -k "concat" -x 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/tests.log
timeout 300 python tools/exp/concat_stats_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/concat_stats_bench.log
B="--steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes"
for rd in 1 2; do
  for opt in "" "--no-gn-epilogue"; do
    timeout 400 python bench.py $B $opt > $O/bench.json 2> $O/bench.err
    python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('round $rd [$opt]', round(d['ms_per_step'],2), [round(p['eps_max_abs_err']*1e4,2) for p in d['parity']['pins']])" | tee -a $O/ab.log
  done
done
