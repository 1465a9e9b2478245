#!/bin/bash
# round 5, call 6: time(K) of classic vs staggered (slope / intercept)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/exp/stagger_ksweep.py > $O/ksweep.log 2>&1
grep -v amdgpu.ids $O/ksweep.log | tail -12
