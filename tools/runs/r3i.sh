#!/bin/bash
# where the ~200 device memcpys per step come from; then the default bench line incl. the CPU oracle leg (final build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i
mkdir -p $O
timeout 300 python tools/exp/find_copies.py tiny 2>&1 | grep -v amdgpu.ids | tail -45 | tee $O/find_copies.log
