#!/bin/bash
# round 4, call 17: whole-step A/B of the GroupNorm statistics from the temporal convs' epilogues against the statistics launches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q
mkdir -p $O
export TMPDIR=/tmp
B="--steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes"
for rd in 1 2 3; do
  for opt in "" "--no-gn-epilogue"; do
    timeout 400 python bench.py $B $opt > $O/bench.json 2> $O/bench.err
    python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('round $rd [$opt]', round(d['ms_per_step'],2), d['parity']['eps_max_abs_err'], d['parity']['eps_mean_abs_err'])" | tee -a $O/ab.log
  done
done
timeout 900 python -m pytest -q --timeout=850 tests/test_model_gpu.py tests/test_view_shard_gpu.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/model_tests.log
