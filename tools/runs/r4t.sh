#!/bin/bash
# round 4, call 20: error / time A/B of the Upsample convs' precise operand (python-level switch; the library is unchanged)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4t
mkdir -p $O
export TMPDIR=/tmp
B="--steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes"
for rd in 1 2; do
  for opt in "" "--upsample-plain-operand"; do
    timeout 300 python bench.py $B $opt > $O/bench.json 2> $O/bench.err
    python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('round $rd [$opt]', round(d['ms_per_step'],2), [round(p['eps_max_abs_err']*1e4,2) for p in d['parity']['pins']], d['parity']['eps_mean_abs_err'])" | tee -a $O/ab.log
  done
done
rm -f gpurun_out/test_measurements.log
PNC_UPSAMPLE_PLAIN=1 timeout 300 python -m pytest -q --timeout=280 tests/test_model_gpu.py -k "full_size_properties_and_golden" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/pins.log
grep "full_cfg" gpurun_out/test_measurements.log | tee -a $O/ab.log
