#!/bin/bash
# round 4, call 2: persistent plain-A GEMM (asm-LDS epilogue): bit-identity tests, kernel A/B, whole-step A/B; attention 42 as default
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q --timeout=900 -x -k "persistent or attn or gemm_plain or layernorm or lo8" 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/kernel_tests.log
timeout 300 python tools/exp/persist_ab.py 3 > $O/persist_ab.log 2>&1
tail -40 $O/persist_ab.log
B="--steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
for rd in 1 2; do
for opt in "" "--set-option GEMM_PERSIST=1" "--set-option ATTN_VARIANT=1"; do
  tag=$(echo "$opt" | tr -c 'A-Za-z0-9=' '_')
  timeout 400 python bench.py $B $opt > $O/bench_${rd}_$tag.json 2> $O/bench_${rd}_$tag.err
  python -c "import json;d=json.loads(open('$O/bench_${rd}_$tag.json').read().strip().splitlines()[-1]);print('round $rd [$opt]', d['ms_per_step'])"
done
done
timeout 400 python bench.py --cpu-baseline none > $O/bench_default_no_cpu.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench_default_no_cpu.json').read().strip().splitlines()[-1]);print('default', d['value'],d['ms_per_step'],d['parity']['eps_max_abs_err'],d['modes']['fast']['ms_per_step'], d['roofline']['frac'])"
timeout 600 python -m pytest tests/test_model_gpu.py -q --timeout=600 -x -k "full_size or golden" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/model_tests.log
