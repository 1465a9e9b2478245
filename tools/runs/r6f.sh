#!/bin/bash
# round 6: whole-step A/B of PNC_OPT_GEMM_STAGGER 4 (new default: level-0 FF1 staggered too) vs 8, same box, interleaved
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f
mkdir -p $O
export TMPDIR=/tmp
COMMON="--steps 10 --warmup 3 --cpu-baseline none --no-modes --no-kernel-breakdown"
for tag in st4 st8 st4b st8b; do
  opt=""; case $tag in st8*) opt="--set-option GEMM_STAGGER=8";; esac
  timeout 400 python bench.py $COMMON $opt > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json;d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]);print('$tag', d['ms_per_step'], d['parity']['eps_max_abs_err'], d['parity']['range_monitor'], d['roofline']['clocks']['sclk_mhz_median'])" || tail -5 $O/bench_$tag.err
done
