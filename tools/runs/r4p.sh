#!/bin/bash
# round 4, call 16: GroupNorm statistics out of the temporal conv's epilogue (PncGemmParams.gn_part): kernel test, model pins, whole-step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest -q --timeout=580 tests/test_kernels_gpu.py -k "groupnorm or conv1d" 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/kernel_tests.log
B="--steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes"
for rd in 1 2; do
  for opt in 1 0; do
    timeout 400 python bench.py $B --set-option GEMM_GN_STATS=$opt > $O/bench_${opt}_${rd}.json 2> $O/bench_${opt}_${rd}.err
    python -c "import json;d=json.loads(open('$O/bench_${opt}_${rd}.json').read().strip().splitlines()[-1]);print('round $rd gn_stats=$opt', round(d['ms_per_step'],2), d['parity']['eps_max_abs_err'])" | tee -a $O/ab.log
  done
done
timeout 900 python -m pytest -q --timeout=850 tests/test_model_gpu.py -k "full_size or golden or frame_shard" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/model_tests.log
