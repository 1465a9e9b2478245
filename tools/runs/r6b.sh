#!/bin/bash
# round 6: the two CFG halves as stream pairs on complementary halves of the CUs (hipExtStreamCreateWithCUMask), unfused step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b
mkdir -p $O
export TMPDIR=/tmp
COMMON="--steps 10 --warmup 3 --cpu-baseline none --no-modes --no-kernel-breakdown --no-fused-step"
run() { tag=$1; shift; timeout 400 python bench.py $COMMON "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err; python -c "import json;d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]);print('$tag', d['ms_per_step'], d['parity']['eps_max_abs_err'], d['roofline'].get('clocks',{}).get('sclk_mhz_median'), d['roofline'].get('clocks',{}).get('package_power_w_mean'))" || tail -5 $O/bench_$tag.err; }
run unfused
run split --split-samples
run split_evenodd --split-samples --cu-split even-odd
run split_nibbles --split-samples --cu-split nibbles
run split_halves --split-samples --cu-split halves
run unfused_again
