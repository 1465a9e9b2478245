#!/bin/bash
# round 5, call 5: whole-network A/B of the specialised staggered schedule (0 vs 8)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/exp/stagger_ab.py 0,8 4 > $O/stagger_ab.log 2>&1
grep -v amdgpu.ids $O/stagger_ab.log | head -40
