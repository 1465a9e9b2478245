#!/bin/bash
# round 5, call 10: the whole GPU suite on the current tree + smoke
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j
mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/test_measurements.log
timeout 2400 python -m pytest tests -m gpu -x -q --timeout=1500 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/gpu_suite.log
cp gpurun_out/test_measurements.log $O/ 2>/dev/null
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.log
