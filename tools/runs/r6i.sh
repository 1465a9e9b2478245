#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i
mkdir -p $O
COMMON="--steps 20 --warmup 3 --cpu-baseline none --no-modes --no-parity"
for tag in text old text_b old_b text_c old_c; do
  opt=""; case $tag in old*) opt="--set-option ATTN_DMA=5";; esac
  timeout 400 python bench.py $COMMON $opt > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json;d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]);k=d['roofline']['kernels'];print('$tag', round(d['ms_per_step'],2), {n:round(v['ms'],2) for n,v in k.items() if 'attn' in n}, d['roofline']['clocks']['sclk_mhz_median'])" || tail -5 $O/bench_$tag.err
done
