#!/bin/bash
# round 6: direct LayerNorm epilogue (epi_direct_ln) — per-shape A/B, the kernel test files, whole GPU suite (eps pins), whole-step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6o
mkdir -p $O
timeout 300 python tools/exp/direct_ln_ab.py 2>&1 | grep -v amdgpu > $O/direct_ln_ab.log; cut -c1-330 $O/direct_ln_ab.log
rm -f gpurun_out/test_measurements.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=1400 -x 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/gpu_tests.log
cp gpurun_out/test_measurements.log $O/ 2>/dev/null
COMMON="--steps 20 --warmup 3 --cpu-baseline none --no-modes"
for tag in new old new_b old_b new_c old_c; do
  opt=""; case $tag in old*) opt="--set-option GEMM_FUSE_LN=5";; esac
  timeout 400 python bench.py $COMMON $opt > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json;d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]);k=d['roofline']['kernels'];print('$tag', round(d['ms_per_step'],2), round(k['gemm_plain']['ms'],2), d['parity']['eps_max_abs_err'], d['roofline']['clocks']['sclk_mhz_median'])" || tail -5 $O/bench_$tag.err
done
