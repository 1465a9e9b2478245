#!/bin/bash
# round 6: direct fp32 epilogue (epi_direct_o32) — per-shape A/Bs, the GEMM / conv test files, whole-step A/B (interleaved)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6n
mkdir -p $O
timeout 300 python tools/exp/direct_epilogue_ab.py 2>&1 | grep -v amdgpu > $O/direct_epilogue_ab.log; cut -c1-220 $O/direct_epilogue_ab.log
timeout 300 python tools/exp/stencil_direct_epilogue_ab.py 2>&1 | grep -v amdgpu > $O/stencil_direct_epilogue_ab.log; cut -c1-200 $O/stencil_direct_epilogue_ab.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_stagger_gpu.py tests/test_lo8_gpu.py tests/test_kernels_vs_oracle_gpu.py -m gpu -q -x 2>&1 | tail -3
COMMON="--steps 20 --warmup 3 --cpu-baseline none --no-modes --no-parity"
for tag in new old new_b old_b new_c old_c; do
  opt=""; case $tag in old*) opt="--set-option GEMM_FUSE_LN=3";; esac
  timeout 400 python bench.py $COMMON $opt > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json;d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]);k=d['roofline']['kernels'];print('$tag', round(d['ms_per_step'],2), {n:round(v['ms'],2) for n,v in k.items() if 'gemm' in n}, d['roofline']['clocks']['sclk_mhz_median'])" || tail -5 $O/bench_$tag.err
done
