#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in new prev new prev; do
  if [ $lib = prev ]; then export PANACEA_HIP_LIB=$GRAFT_REPO_ROOT/panacea_amd/lib/exp/libpanacea_hip_ptr.so; else unset PANACEA_HIP_LIB; fi
  echo "== $lib"; timeout 200 python tools/kbench.py halo "L2 conv3x3" 2>&1 | grep "halo"
done | tee gpurun_out/r2y_l2_bn256.log
