#!/bin/bash
# round 3, third GPU pass: the 128x320 two-workgroups-per-CU geometry (per-shape A/B, whole-step A/B) and the Infinity-Cache row
# panels of GroupNorm / FeedForward (whole-step A/B), interleaved on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/runs/r3c.py > $O/kbench_two_wg.log 2>&1
cat $O/kbench_two_wg.log | grep -v amdgpu.ids
B="python bench.py --steps 6 --warmup 2 --cpu-baseline none --no-modes --no-kernel-breakdown"
for rep in 1 2; do
for v in "base --two-wg 0 --mall-panels 0" "twowg --two-wg 1 --mall-panels 0" "mall --two-wg 0 --mall-panels 112" "both --two-wg 1 --mall-panels 112"; do
  set -- $v; n=$1; shift
  timeout 300 $B "$@" > $O/bench_${n}_$rep.json 2> $O/bench_${n}_$rep.err
  python -c "
import json; d=json.loads(open('$O/bench_${n}_$rep.json').read().strip().splitlines()[-1]); print('$n', $rep, round(d['ms_per_step'],2), d['parity']['eps_max_abs_err'])"
done; done
