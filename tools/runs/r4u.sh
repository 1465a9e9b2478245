#!/bin/bash
# round 4, call 21 (last GPU minutes): GroupNorm records from the STT proj_out GEMMs — kernel tests, the traffic stamp and the default
# bench line of this library, then the model pins and the A/B against a statistics launch per GroupNorm
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4u
mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/test_measurements.log
timeout 200 python -m pytest -q --timeout=190 -x tests/test_kernels_gpu.py -k "groupnorm_records" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/kernel_tests.log
grep -q "passed" $O/kernel_tests.log && ! grep -q "failed" $O/kernel_tests.log || exit 1
ARGS="--steps 1 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
BENCH1="python $GRAFT_REPO_ROOT/bench.py $ARGS"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- $BENCH1 > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1)
done
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 2 precise "bench.py $ARGS" > $O/pmc.log 2>&1
mkdir -p $O/pmc && cp profiles/round4/pmc_* $O/pmc/ 2>/dev/null
head -13 $O/pmc.log | tail -3
timeout 200 python bench.py --cpu-baseline none > $O/bench_default_no_cpu.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench_default_no_cpu.json').read().strip().splitlines()[-1]);print('default', d['value'],d['ms_per_step'],[p['eps_max_abs_err'] for p in d['parity']['pins']],d['modes']['fast']['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))" | tee $O/default.log
timeout 120 python bench.py --steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity --no-gn-epilogue > $O/bench_nogn.json 2>/dev/null
python -c "import json;d=json.loads(open('$O/bench_nogn.json').read().strip().splitlines()[-1]);print('no-gn-epilogue', d['ms_per_step'])" | tee -a $O/default.log
timeout 120 python bench.py --steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity > $O/bench_gn.json 2>/dev/null
python -c "import json;d=json.loads(open('$O/bench_gn.json').read().strip().splitlines()[-1]);print('gn-epilogue', d['ms_per_step'])" | tee -a $O/default.log
timeout 200 python -m pytest -q --timeout=190 -x tests/test_model_gpu.py -k "full_size_properties_and_golden" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/model_tests.log
cp gpurun_out/test_measurements.log $O/ 2>/dev/null
