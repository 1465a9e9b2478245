#!/bin/bash
# micro-overheads of the GEMM main loop: this build vs the previous one (same box), kernel tests first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo skip tests
for lib in new prev new prev; do
  if [ $lib = prev ]; then export PANACEA_HIP_LIB=$GRAFT_REPO_ROOT/panacea_amd/lib/exp/libpanacea_hip_ptr.so; else unset PANACEA_HIP_LIB; fi
  timeout 300 python bench.py --steps 5 --warmup 2 --cpu-baseline none --no-modes 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$lib', round(d['ms_per_step'],2), d['parity']['eps_max_abs_err'], ' '.join(f\"{n}={v['ms']:.2f}\" for n,v in k.items() if v['ms']>1.5))"
done 2>&1 | tee gpurun_out/r2u_family_ab.log
