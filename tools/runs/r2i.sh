#!/bin/bash
# grouped tile order: bit-identity tests, per-shape A/B, whole-step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout=600 -k "gemm" 2>&1 | tail -6 > gpurun_out/r2i_pytest.log
tail -3 gpurun_out/r2i_pytest.log
timeout 500 python tools/kbench.py group_m > gpurun_out/r2i_group_m.log 2>&1
cat gpurun_out/r2i_group_m.log | tail -20
for rep in 1 2; do
  for g in 0 1; do
    timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline none --no-kernel-breakdown --no-modes --gemm-group-m $g > gpurun_out/r2i_bench_g${g}_$rep.json 2> gpurun_out/r2i_bench_g${g}_$rep.err
    python -c "import json;d=json.loads(open('gpurun_out/r2i_bench_g${g}_$rep.json').read().strip().splitlines()[-1]);print('group_m $g rep $rep', round(d['ms_per_step'],2), d['parity']['eps_max_abs_err'])"
  done
done
