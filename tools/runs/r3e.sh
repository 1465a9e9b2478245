#!/bin/bash
# ablation probe of the fused feed-forward kernel (where does its time go)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
timeout 120 tools/exp/ffchain_probe 2>&1 | tee gpurun_out/r3e/ffchain_probe.log
