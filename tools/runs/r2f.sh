#!/bin/bash
# sixth GPU pass: fused LayerNorm with on-chip values — kernel + model tests, in-process A/B (fusion on / off)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -k "layernorm or golden or full or fused or text or determin or properties" 2>&1 | tail -15 > gpurun_out/r2f_pytest.log
tail -5 gpurun_out/r2f_pytest.log
for rep in 1 2; do
  for f in on off; do
    fl=""; if [ $f = off ]; then fl="--no-ln-fusion"; fi
    timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline none --no-kernel-breakdown --no-modes $fl > gpurun_out/r2f_bench_${f}_$rep.json 2> gpurun_out/r2f_bench_${f}_$rep.err
    python -c "import json;d=json.loads(open('gpurun_out/r2f_bench_${f}_$rep.json').read().strip().splitlines()[-1]);print('ln fusion $f $rep', round(d['ms_per_step'],2), d['parity']['eps_max_abs_err'])"
  done
done
timeout 200 python tools/kbench.py "L0 proj" > gpurun_out/r2f_kbench.log 2>&1; grep "L0 proj" gpurun_out/r2f_kbench.log
