#!/bin/bash
# seventh GPU pass: cost / error points of the operand policies, first-stage benches on the round-2 kernels, default bench
# with the whole-step CPU oracle leg
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for prec in precise-lite precise-all; do
  timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline none --no-kernel-breakdown --no-modes --precision $prec > gpurun_out/r2g_bench_$prec.json 2> gpurun_out/r2g_bench_$prec.err
  python -c "import json;d=json.loads(open('gpurun_out/r2g_bench_$prec.json').read().strip().splitlines()[-1]);print('$prec', round(d['ms_per_step'],2), d['parity']['eps_max_abs_err'], d['parity']['eps_mean_abs_err'])"
done
timeout 300 python bench.py --stage vae-decode --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2g_vae_decode.json 2>/dev/null
timeout 300 python bench.py --stage vae-encode --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2g_vae_encode.json 2>/dev/null
python -c "
import json
for n in ('decode','encode'):
    d=json.loads(open('gpurun_out/r2g_vae_%s.json'%n).read().strip().splitlines()[-1]); print(n, round(d['ms_per_step'],2), round(d['roofline']['frac'],3))"
timeout 1500 python bench.py > gpurun_out/r2g_bench_default.json 2> gpurun_out/r2g_bench_default.err
tail -3 gpurun_out/r2g_bench_default.err
python -c "import json;d=json.loads(open('gpurun_out/r2g_bench_default.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['cpu_baseline'])"
