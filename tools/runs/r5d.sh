#!/bin/bash
# round 5, call 4: the staggered loop specialised to plain A with a branch-free DMA issue — bit identity, isolated A/B, tail16 / w1 pins
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest -q --timeout=500 tests/test_gemm_stagger_gpu.py -x 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/tests.log
timeout 900 python tools/exp/stagger_kbench.py "L" > $O/stagger_kbench.log 2>&1
grep -v amdgpu.ids $O/stagger_kbench.log | grep -v "conv" | tail -20
rm -f gpurun_out/test_measurements.log
timeout 900 python -m pytest -q --timeout=800 tests/test_model_gpu.py -k "other_weight_sets" -s 2>&1 | grep -v amdgpu.ids | grep -E "vs reference|passed|failed|Error|error|assert" | tail -8 | tee $O/pins.log
