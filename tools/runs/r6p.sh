#!/bin/bash
# round 6: FF1 (persistent GEGLU kernel) epilogue with two-byte buffer stores straight from the registers (PNC_OPT_GEMM_STAGGER + 256,
# experiment) against the 2 KB LDS slab per wave: whole-step A/B (interleaved)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p
mkdir -p $O
COMMON="--steps 20 --warmup 3 --cpu-baseline none --no-modes --no-parity"
for tag in new old new_b old_b new_c old_c; do
  opt=""; case $tag in old*) opt="--set-option GEMM_STAGGER=260";; esac
  timeout 400 python bench.py $COMMON $opt > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json;d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]);k=d['roofline']['kernels'];print('$tag', round(d['ms_per_step'],2), round(k['gemm_plain']['ms'],2), d['roofline']['clocks']['sclk_mhz_median'])" || tail -5 $O/bench_$tag.err
done
