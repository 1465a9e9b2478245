#!/bin/bash
# stencil-tile kernels (conv3x3 + conv1d-T): parity (bit-identity with the per-tap gather), kernel A/B, step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=300 -k "conv" 2>&1 | tail -15 > gpurun_out/r2k_pytest.log
cat gpurun_out/r2k_pytest.log
timeout 600 python tools/kbench.py halo > gpurun_out/r2k_kbench_halo.log 2>&1
cat gpurun_out/r2k_kbench_halo.log
for h in 0 1 2 0 1 2; do
  timeout 600 python bench.py --steps 6 --warmup 2 --cpu-baseline none --no-modes --no-kernel-breakdown --stencil-tiles $h 2>gpurun_out/r2k_bench_h$h.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiles=$h', d['ms_per_step'], d['parity']['eps_max_abs_err'])"
done | tee gpurun_out/r2k_bench_ab.log
