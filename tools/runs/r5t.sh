#!/bin/bash
# round 5: emb_layers of every ResBlock in one launch per network (pnc_linear_smallm_segments): test, pins, A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5t
mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "smallm" --timeout=250 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/tests.log
timeout 600 python -m pytest tests/test_model_gpu.py -q -x --timeout=550 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/tests.log
for i in 1 2; do
  for v in 0 1; do
    PNC_EMB_BATCH=$v timeout 300 python bench.py --steps 20 --warmup 4 --cpu-baseline none --no-modes --no-kernel-breakdown --no-parity > $O/b_${v}_${i}.json 2>> $O/bench.err
    python -c "import json;d=json.loads(open('$O/b_${v}_${i}.json').read().strip().splitlines()[-1]);print('PNC_EMB_BATCH=$v', round(d['ms_per_step'],2))" | tee -a $O/ab.log
  done
done
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-baseline none --no-modes > $O/bench.json 2>> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['parity']['eps_max_abs_err'], d['roofline']['kernels'].get('linear_smallm'))" | tee -a $O/ab.log
