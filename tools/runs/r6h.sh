#!/bin/bash
# round 6: dedicated few-key (text) attention kernel: tests + whole-step A/B (ATTN_VARIANT=42 keeps attn_views_kernel for the text launches; it is also the default for the view launches)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attn" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/attn_tests.log
COMMON="--steps 10 --warmup 3 --cpu-baseline none --no-modes --no-kernel-breakdown"
for tag in text old text_b old_b; do
  opt=""; case $tag in old*) opt="--set-option ATTN_DMA=5";; esac
  timeout 400 python bench.py $COMMON $opt > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json;d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]);print('$tag', d['ms_per_step'], d['parity']['eps_max_abs_err'], [p['eps_max_abs_err'] for p in d['parity']['pins']], d['roofline']['clocks']['sclk_mhz_median'])" || tail -5 $O/bench_$tag.err
done
