#!/bin/bash
# fourth GPU pass: GEMM + conditioner + sampler tests on the new build, same-box A/B of the residual-prefetch depth
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -k "gemm or fused or text_tower or pipeline or softmax or vae" 2>&1 | tail -12 > gpurun_out/r2d_pytest.log
tail -4 gpurun_out/r2d_pytest.log
for rep in 1 2; do
  for lib in new base; do
    if [ $lib = base ]; then export PANACEA_HIP_LIB=$GRAFT_REPO_ROOT/panacea_amd/lib/libpanacea_hip_base.so; else unset PANACEA_HIP_LIB; fi
    timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline none --no-kernel-breakdown > gpurun_out/r2d_bench_${lib}_$rep.json 2> gpurun_out/r2d_bench_${lib}_$rep.err
    python -c "import json;d=json.loads(open('gpurun_out/r2d_bench_${lib}_$rep.json').read().strip().splitlines()[-1]);print('$lib $rep', round(d['ms_per_step'],2), round(d['modes']['fast']['ms_per_step'],2), d['parity']['eps_max_abs_err'])"
  done
done
unset PANACEA_HIP_LIB
timeout 200 python tools/kbench.py "proj" > gpurun_out/r2d_kbench_new.log 2>&1
PANACEA_HIP_LIB=$GRAFT_REPO_ROOT/panacea_amd/lib/libpanacea_hip_base.so timeout 200 python tools/kbench.py "proj" > gpurun_out/r2d_kbench_base.log 2>&1
grep -h "L0 proj\|L1 proj" gpurun_out/r2d_kbench_new.log gpurun_out/r2d_kbench_base.log
timeout 200 python bench.py --steps 10 --warmup 2 --cpu-baseline none --no-kernel-breakdown --no-modes --split-samples > gpurun_out/r2d_bench_split.json 2>/dev/null
python -c "import json;d=json.loads(open('gpurun_out/r2d_bench_split.json').read().strip().splitlines()[-1]);print('split-samples', round(d['ms_per_step'],2))"
timeout 200 python bench.py --steps 10 --warmup 2 --cpu-baseline none --no-kernel-breakdown --no-modes --hoist > gpurun_out/r2d_bench_hoist.json 2>/dev/null
python -c "import json;d=json.loads(open('gpurun_out/r2d_bench_hoist.json').read().strip().splitlines()[-1]);print('hoist', round(d['ms_per_step'],2))"
timeout 200 python bench.py --steps 10 --warmup 2 --cpu-baseline none --no-modes --frames 1 > gpurun_out/r2d_bench_frames1.json 2>/dev/null
python -c "import json;d=json.loads(open('gpurun_out/r2d_bench_frames1.json').read().strip().splitlines()[-1]);print('frames1', round(d['ms_per_step'],2), d['roofline']['frac'])"
timeout 300 python bench.py --yaml-exact --steps 25 --warmup 2 --cpu-baseline none --no-modes --no-kernel-breakdown > gpurun_out/r2d_bench_yaml.json 2>/dev/null
python -c "import json;d=json.loads(open('gpurun_out/r2d_bench_yaml.json').read().strip().splitlines()[-1]);print('yaml-exact', round(d['ms_per_step'],2), d['config']['workload'][:60])"
