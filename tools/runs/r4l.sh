#!/bin/bash
# round 4, call 12: view-band convs read the neighbours' columns in place (PncGemmParams.x_halo_off): kernel tests, view-shard tests, step time
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest -q --timeout=850 tests/test_kernels_gpu.py -k "conv3x3 or conv" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/kernel_tests.log
timeout 900 python -m pytest -q --timeout=850 tests/test_view_shard_gpu.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/view_tests.log
B="--steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
timeout 400 python bench.py $B > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('step', d['ms_per_step'])" | tee $O/step.log
timeout 400 python tools/exp/view_loopback_time.py 5 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/view_loopback_time.txt
