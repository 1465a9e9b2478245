#!/bin/bash
# round 4, call 4: LayerNorm variants of the persistent plain-A GEMM (variance FMA written out), whole-step A/B of the two new defaults
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest -q --timeout=280 "tests/test_kernels_gpu.py::test_gemm_persistent_kernel_is_bit_identical" "tests/test_kernels_gpu.py::test_gemm_fused_layernorm" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/persist_tests.log
B="--steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
for rd in 1 2; do
for opt in "" "--set-option GEMM_PERSIST=1" "--set-option ATTN_VARIANT=1"; do
  tag=$(echo "$opt" | tr -c 'A-Za-z0-9=' '_')
  timeout 400 python bench.py $B $opt > $O/bench_${rd}_$tag.json 2> $O/bench_${rd}_$tag.err
  python -c "import json;d=json.loads(open('$O/bench_${rd}_$tag.json').read().strip().splitlines()[-1]);print('round $rd [$opt]', d['ms_per_step'])" | tee -a $O/ab.log
done
done
