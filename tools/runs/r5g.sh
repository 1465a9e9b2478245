#!/bin/bash
# round 5, call 7: final stagger policy (GEGLU persistent only): thresholds 0 / 8 / 4 in the network; bit-identity tests; the heavy-tail pin (gain 64);
# GroupNorm records from the concat (test + whole-network A/B through engine.GN_FROM_EPILOGUE is not separable: timed against r5e's numbers)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest -q --timeout=500 tests/test_gemm_stagger_gpu.py tests/test_kernels_gpu.py -k "stagger or concat or layout_helpers" -x 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/tests.log
timeout 900 python tools/exp/stagger_ab.py 0,8,4 4 > $O/stagger_ab.log 2>&1
grep -v amdgpu.ids $O/stagger_ab.log | head -12
rm -f gpurun_out/test_measurements.log
timeout 900 python -m pytest -q --timeout=800 tests/test_model_gpu.py -k "other_weight_sets or full_size_properties" -s 2>&1 | grep -v amdgpu.ids | grep -E "vs reference|passed|failed|Error|error|assert" | tail -12 | tee $O/pins.log
cp gpurun_out/test_measurements.log $O/ 2>/dev/null
