#!/bin/bash
# operand DMA through buffer resources (this build) vs per-lane 64-bit pointers (previous commit's build): kernel tests, kernel and step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=300 -k "gemm" -x 2>&1 | tail -8 | tee gpurun_out/r2s_pytest.log
for rep in 1 2; do
  for lib in buf ptr; do
    if [ $lib = ptr ]; then export PANACEA_HIP_LIB=$GRAFT_REPO_ROOT/panacea_amd/lib/exp/libpanacea_hip_ptr.so; else unset PANACEA_HIP_LIB; fi
    echo "== $lib (rep $rep)"
    if [ $rep = 1 ]; then timeout 300 python tools/kbench.py "gemm L" 2>&1 | grep -v "amdgpu\|Radeon"; fi
    timeout 300 python bench.py --steps 5 --warmup 2 --cpu-baseline none --no-modes --no-kernel-breakdown 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['parity']['eps_max_abs_err'])"
  done
done 2>&1 | tee gpurun_out/r2s_buf_ab.log
