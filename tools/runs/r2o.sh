#!/bin/bash
# A/B of two builds of the same sources: default flags vs -fno-slp-vectorize (the GEGLU epilogue's packed-f32 ops + moves)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import torch" >/dev/null 2>&1
for rep in 1 2; do
  for lib in default noslp; do
    if [ $lib = noslp ]; then export PANACEA_HIP_LIB=$GRAFT_REPO_ROOT/panacea_amd/lib/exp/libpanacea_hip_noslp.so; else unset PANACEA_HIP_LIB; fi
    echo "== $lib (rep $rep)"
    timeout 300 python tools/kbench.py "ff1" 2>&1 | grep -v "amdgpu\|Radeon"
    timeout 300 python tools/kbench.py "attn L0" 2>&1 | grep "intra\|cross"
    timeout 300 python bench.py --steps 5 --warmup 2 --cpu-baseline none --no-modes --no-kernel-breakdown 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['parity']['eps_max_abs_err'])"
  done
done 2>&1 | tee gpurun_out/r2o_noslp_ab.log
