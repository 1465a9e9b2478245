#!/bin/bash
# round 4, call 7: XCD-aware workgroup mapping of the attention kernel (tests, kernel A/B table, whole step, FETCH_SIZE pass);
# sharded temporal GroupNorm parts + halo-layout temporal conv (kernel tests, loop-back network tests)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest -q --timeout=580 tests/test_kernels_gpu.py -k "attn or conv1d or groupnorm_temporal" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/kernel_tests.log
timeout 300 python tools/exp/attn_ab.py 2 > $O/attn_ab.log 2>&1
grep "L0\|L1" $O/attn_ab.log | grep "round 1" | grep "82 defer8 inc\|42 defer8 inc"
B="--steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
for rd in 1 2; do
for opt in "" "--set-option ATTN_VARIANT=1"; do
  tag=$(echo "$opt" | tr -c 'A-Za-z0-9=' '_')
  timeout 400 python bench.py $B $opt > $O/bench_${rd}_$tag.json 2> $O/bench_${rd}_$tag.err
  python -c "import json;d=json.loads(open('$O/bench_${rd}_$tag.json').read().strip().splitlines()[-1]);print('round $rd [$opt]', d['ms_per_step'])" | tee -a $O/ab.log
done
done
ARGS="--steps 1 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_F -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/$O/pmc_F.log 2>&1)
python - <<'PY' | tee $O/fetch_attn.txt
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob('/tmp/pmc_F/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE':
            agg[r['Kernel_Name'][:80]] += float(r['Counter_Value'])
tot = sum(agg.values()) * 1024 * 2 / 2 / 1e9
print('fetch total GB/eval (x2 factor)', round(tot, 1))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:6]:
    print(f'{k:80s} {v * 1024 * 2 / 2 / 1e9:8.2f} GB/eval')
PY
timeout 900 python -m pytest -q --timeout=880 tests/test_model_gpu.py tests/test_view_shard_gpu.py -k "frame_shard or loop_back or full_size" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/model_tests.log
