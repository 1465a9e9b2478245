#!/bin/bash
# round 4, call 5: causal attention for the text tower (kernel test, tower vs oracle, --stage text-tower timing)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e
mkdir -p $O
timeout 400 python -m pytest -q --timeout=380 tests/test_kernels_gpu.py -k "attn" tests/test_conditioner.py -m gpu 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests.log
timeout 300 python bench.py --stage text-tower --steps 20 --warmup 3 > $O/text_tower.json 2> $O/text_tower.err
tail -1 $O/text_tower.json
