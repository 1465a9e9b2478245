#!/bin/bash
# round 4, call 8: frame-fastest tile order of the temporal conv (tests, per-shape profile, whole step, FETCH_SIZE pass)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest -q --timeout=580 tests/test_kernels_gpu.py -k "conv1d" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/kernel_tests.log
B="--steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
for rd in 1 2; do
  timeout 400 python bench.py $B > $O/bench_${rd}.json 2> $O/bench_${rd}.err
  python -c "import json;d=json.loads(open('$O/bench_${rd}.json').read().strip().splitlines()[-1]);print('round $rd', d['ms_per_step'])" | tee -a $O/ab.log
done
timeout 300 python tools/shape_profile.py precise 2>&1 | grep -v amdgpu.ids > $O/shape_profile_precise.log; grep "mode2" $O/shape_profile_precise.log
ARGS="--steps 1 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_F -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/$O/pmc_F.log 2>&1)
python - <<'PY' | tee $O/fetch.txt
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob('/tmp/pmc_F/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE':
            agg[r['Kernel_Name'][:80]] += float(r['Counter_Value'])
tot = sum(agg.values()) * 1024 * 2 / 2 / 1e9
print('fetch total GB/eval (x2 factor)', round(tot, 1))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:24]:
    print(f'{k:80s} {v * 1024 * 2 / 2 / 1e9:8.2f} GB/eval')
PY
