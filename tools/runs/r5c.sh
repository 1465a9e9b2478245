#!/bin/bash
# round 5, call 3: why the staggered schedule wins in the probe and loses in the network — probe with DMA placement variants, warm vs
# cold operands; the library's own launches A/B'd in isolation (warm / cold); the staggered stencil-tile kernel (bit identity + A/B)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c
mkdir -p $O
export TMPDIR=/tmp
timeout 300 tools/exp/gemm_phase_probe 5 > $O/gemm_phase_probe.log 2>&1; echo "probe rc $?" >> $O/gemm_phase_probe.log
grep -E "warm|cold|FAIL|rc" $O/gemm_phase_probe.log | cut -c1-400
timeout 600 python -m pytest -q --timeout=500 tests/test_gemm_stagger_gpu.py tests/test_kernels_gpu.py -k "stagger or stencil" -x 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/tests.log
timeout 900 python tools/exp/stagger_kbench.py > $O/stagger_kbench.log 2>&1
grep -v amdgpu.ids $O/stagger_kbench.log | tail -20
