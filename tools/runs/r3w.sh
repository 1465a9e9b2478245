#!/bin/bash
# persistent GEGLU GEMM (opt-in): bit-identity test, per-shape A/B, whole-step A/B through bench.py --gemm-persist
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3w
mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=200 -k "persistent or geglu" 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/tests.log
timeout 200 python tools/exp/persist_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/persist_ab.log
