#!/bin/bash
# round-3 state check: the whole -m gpu suite (new: two-process view-shard test, ViT-H/14 text tower, network with the fused
# feed-forward), smoke(), bench of the default mode without the CPU leg, text-tower stage
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
rm -f gpurun_out/test_measurements.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -v amdgpu.ids | tail -25 | tee gpurun_out/r3f/gpu_tests.log
cp gpurun_out/test_measurements.log gpurun_out/r3f/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/r3f/smoke.log
timeout 400 python bench.py --cpu-baseline none 2>gpurun_out/r3f/bench.err | tail -1 > gpurun_out/r3f/bench_default.json
timeout 200 python bench.py --stage text-tower 2>gpurun_out/r3f/tt.err | tail -1 > gpurun_out/r3f/bench_text_tower.json
tail -n 3 gpurun_out/r3f/bench.err; tail -n 3 gpurun_out/r3f/tt.err
head -c 600 gpurun_out/r3f/bench_default.json; echo; cat gpurun_out/r3f/bench_text_tower.json
