#!/bin/bash
# round 3, third GPU pass: the 128x320 two-workgroups-per-CU geometry — bit-identity test, per-shape A/B, whole-step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q --timeout=600 -k "two_workgroups or fused_layernorm or precise_operand_plain or e4m3" 2>&1 | tail -15 > $O/tests.log
tail -3 $O/tests.log
timeout 600 python tools/runs/r3c.py > $O/kbench_two_wg.log 2>&1
cat $O/kbench_two_wg.log | grep -v amdgpu.ids
