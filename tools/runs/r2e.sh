#!/bin/bash
# fifth GPU pass: fused LayerNorm — kernel tests, full model tests, in-process A/B (fusion on / off), rocprof launch counts
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 2>&1 | tail -15 > gpurun_out/r2e_pytest.log
tail -5 gpurun_out/r2e_pytest.log
for rep in 1 2; do
  for f in on off; do
    fl=""; if [ $f = off ]; then fl="--no-ln-fusion"; fi
    timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline none --no-kernel-breakdown --no-modes $fl > gpurun_out/r2e_bench_${f}_$rep.json 2> gpurun_out/r2e_bench_${f}_$rep.err
    python -c "import json;d=json.loads(open('gpurun_out/r2e_bench_${f}_$rep.json').read().strip().splitlines()[-1]);print('ln fusion $f $rep', round(d['ms_per_step'],2), d['parity']['eps_max_abs_err'])"
  done
done
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2e_prof -- $BENCH > $GRAFT_REPO_ROOT/gpurun_out/r2e_prof.log 2>&1)
find gpurun_out/r2e_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2e_kernel_stats.csv
rm -rf gpurun_out/r2e_prof
grep -c layernorm gpurun_out/r2e_kernel_stats.csv; grep layernorm gpurun_out/r2e_kernel_stats.csv | cut -c1-120
