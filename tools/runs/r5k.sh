#!/bin/bash
# round 5, call 12: split K with 4 slices from 56 K tiles on (level-3 temporal convs / FF2): shape profile + bench; --split-samples A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest -q --timeout=500 tests/test_kernels_gpu.py -k "splitk" -x 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/tests.log
timeout 300 python tools/shape_profile.py precise 2>&1 | grep -v amdgpu.ids > $O/shape_profile_precise.log; head -2 $O/shape_profile_precise.log; grep "M=   3072" $O/shape_profile_precise.log | head -8
B="--steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes"
for rd in 1 2; do
  for opt in "" "--split-samples"; do
    timeout 400 python bench.py $B $opt > $O/bench.json 2> $O/bench.err
    python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('round $rd [$opt]', round(d['ms_per_step'],2), [round(p['eps_max_abs_err']*1e4,2) for p in d['parity']['pins']])" | tee -a $O/ab.log
  done
done
