#!/bin/bash
# round 6: sum-triggered running max of the view attention: kernel A/B, its tests, whole-step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/exp/attn_sumtrig_ab.py 3 2>&1 | grep -v amdgpu.ids | tee $O/attn_sumtrig_ab.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attn" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/attn_tests.log
COMMON="--steps 10 --warmup 3 --cpu-baseline none --no-modes --no-kernel-breakdown"
for tag in trig12 trig0 trig12b trig0b; do
  opt=""; case $tag in trig0*) opt="--set-option ATTN_SUM_TRIGGER=0";; esac
  timeout 400 python bench.py $COMMON $opt > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json;d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]);print('$tag', d['ms_per_step'], d['parity']['eps_max_abs_err'], d['roofline']['clocks']['sclk_mhz_median'])" || tail -5 $O/bench_$tag.err
done
