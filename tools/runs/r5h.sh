#!/bin/bash
# round 5, call 8: halo-view attention + ControlNet side stream over its own groups (loop-back tests, loop-back timing), concat-with-records micro-benchmark
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest -q --timeout=1400 tests/test_view_shard_gpu.py -x 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/view_tests.log
timeout 300 python tools/exp/concat_stats_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/concat_stats_bench.log
timeout 900 python tools/exp/view_loopback_time.py 5 2>&1 | grep -v amdgpu.ids | tee $O/view_loopback_time.log
timeout 600 python -m pytest -q --timeout=500 tests/test_kernels_gpu.py -k "attn" -x 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/attn_tests.log
