#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=300 -k "stencil" 2>&1 | tail -6 | tee gpurun_out/r2r_pytest.log
timeout 300 python tools/kbench.py halo "conv3x3" 2>&1 | grep -v "amdgpu\|Radeon" | tee gpurun_out/r2r_kbench.log
