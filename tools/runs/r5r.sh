#!/bin/bash
# round 5: text K/V of a network as ONE GEMM (K row-major | V channel-major via n_split) vs one per width: pins, A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r
mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py -q -x --timeout=550 -k "golden or properties" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests.log
for i in 1 2; do
  for v in 0 1; do
    PNC_TEXTKV_ONE_GEMM=$v timeout 300 python bench.py --steps 20 --warmup 4 --cpu-baseline none --no-modes --no-kernel-breakdown > $O/b_${v}_${i}.json 2>> $O/bench.err
    python -c "import json;d=json.loads(open('$O/b_${v}_${i}.json').read().strip().splitlines()[-1]);print('PNC_TEXTKV_ONE_GEMM=$v', round(d['ms_per_step'],2), d['parity']['eps_max_abs_err'], d['roofline']['clocks']['sclk_mhz_median'])" | tee -a $O/ab.log
  done
done
