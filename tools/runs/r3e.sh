#!/bin/bash
# fused feed-forward kernel: correctness tests, ablation probe, level-0 timing against the launch sequence it replaces
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
timeout 300 python -m pytest tests/test_ff_chain_gpu.py -q --timeout=120 -s 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/r3e/ffchain_tests.log
timeout 120 tools/exp/ffchain_probe 2>&1 | tee gpurun_out/r3e/ffchain_probe.log
timeout 200 python tools/runs/r3d_ffchain.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3e/ffchain_kbench.log
