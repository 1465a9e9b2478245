#!/bin/bash
# round 5, call 2: the staggered GEMM schedule in the product — bit-identity tests, whole-network A/B (option 0 / 8 / 1) with the per-shape table
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest -q --timeout=600 tests/test_gemm_stagger_gpu.py -x 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/stagger_tests.log
timeout 900 python tools/exp/stagger_ab.py 0,8,1 3 > $O/stagger_ab.log 2>&1
grep -v amdgpu.ids $O/stagger_ab.log | head -70
