#!/bin/bash
# round 6: whole -m gpu suite on the current library (range monitor, sum-triggered max, acceptance trajectory, full-eps pins)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6d
mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/test_measurements.log
timeout 2400 python -m pytest tests -m gpu -q --timeout=2000 -x 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/gpu_tests.log
cp gpurun_out/test_measurements.log $O/ 2>/dev/null
