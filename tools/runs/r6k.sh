#!/bin/bash
# round 6: stencil_tile_kernel fragment addresses one batch ahead (and, first version, the DMA order) — per-shape A/B on the rebuilt library, the stencil bit-identity tests, whole-step A/B (interleaved)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6m
mkdir -p $O
timeout 300 python tools/exp/stencil_dma_late_ab.py 2>&1 | grep -v amdgpu > $O/stencil_dma_late_ab.log; cat $O/stencil_dma_late_ab.log
timeout 600 python -m pytest tests -m gpu -q -x -k "stencil or conv3x3 or tile" 2>&1 | tail -3
COMMON="--steps 20 --warmup 3 --cpu-baseline none --no-modes --no-parity"
for tag in new old new_b old_b new_c old_c; do
  opt=""; case $tag in old*) opt="--set-option STENCIL_TILES=5";; esac
  timeout 400 python bench.py $COMMON $opt > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json;d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]);k=d['roofline']['kernels'];print('$tag', round(d['ms_per_step'],2), {n:round(v['ms'],2) for n,v in k.items() if 'conv3' in n or 'stencil' in n}, d['roofline']['clocks']['sclk_mhz_median'])" || tail -5 $O/bench_$tag.err
done
