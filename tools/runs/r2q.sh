#!/bin/bash
# stencil tiles with the tail split: parity, kernel A/B at the level-1 shapes, step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=300 -k "conv3x3" 2>&1 | tail -6 | tee gpurun_out/r2q_pytest.log
timeout 300 python tools/kbench.py halo "conv3x3" 2>&1 | grep -v "amdgpu\|Radeon" | tee gpurun_out/r2q_kbench.log
for h in 0 1 0 1; do
  timeout 600 python bench.py --steps 6 --warmup 2 --cpu-baseline none --no-modes --no-kernel-breakdown --stencil-tiles $h 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiles=$h', d['ms_per_step'], d['parity']['eps_max_abs_err'])"
done | tee gpurun_out/r2q_bench_ab.log
