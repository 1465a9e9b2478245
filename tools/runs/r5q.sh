#!/bin/bash
# round 5: the whole -m gpu suite on the final tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q
mkdir -p $O
rm -f gpurun_out/test_measurements.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=1400 -x 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/gpu_tests.log
cp gpurun_out/test_measurements.log $O/ 2>/dev/null
