#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in buf ptr buf ptr; do
  if [ $lib = ptr ]; then export PANACEA_HIP_LIB=$GRAFT_REPO_ROOT/panacea_amd/lib/exp/libpanacea_hip_ptr.so; else unset PANACEA_HIP_LIB; fi
  timeout 300 python bench.py --steps 5 --warmup 2 --cpu-baseline none --no-modes 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$lib', round(d['ms_per_step'],2), ' '.join(f\"{n}={v['ms']:.2f}\" for n,v in k.items() if v['ms']>1.5))"
done 2>&1 | tee gpurun_out/r2t_family_ab.log
