#!/bin/bash
# round 6: kernel stats (one-stream) + per-shape profile of the current library
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6g
mkdir -p $O
export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r6g_prof -- $BENCH > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
find /tmp/r6g_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
timeout 300 python tools/shape_profile.py precise 2>&1 | grep -v amdgpu.ids > $O/shape_profile_precise.log; head -3 $O/shape_profile_precise.log
