#!/bin/bash
# round 6, checkpoint measurement of the shipped library: PMC passes (traffic + MFMA busy, both stamped), default bench incl. the CPU
# leg with its thread sweep, whole -m gpu suite, rocprofv3 kernel stats, per-shape profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6z
mkdir -p $O
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
BENCH1="python $GRAFT_REPO_ROOT/bench.py $ARGS"
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- $BENCH1 > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1)
done
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 2 precise "bench.py $ARGS" > $O/pmc.log 2>&1
python tools/pmc_traffic.py --mfma /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES 2 precise 165.0 >> $O/pmc.log 2>&1
mkdir -p $O/pmc && cp profiles/round6/pmc_* $O/pmc/ 2>/dev/null
head -13 $O/pmc.log | tail -3
timeout 400 python bench.py --cpu-baseline none > $O/bench_default_no_cpu.json 2> $O/bench_no_cpu.err
python -c "import json;d=json.loads(open('$O/bench_default_no_cpu.json').read().strip().splitlines()[-1]);print('default(no cpu leg)', d['value'],d['ms_per_step'],d['parity']['eps_max_abs_err'],d['modes']['fast']['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('mfma_busy'))"
rm -f gpurun_out/test_measurements.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=1400 -x 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/gpu_tests.log
cp gpurun_out/test_measurements.log $O/ 2>/dev/null
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r6z_prof -- $BENCH > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
find /tmp/r6z_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
timeout 300 python tools/shape_profile.py precise 2>&1 | grep -v amdgpu.ids > $O/shape_profile_precise.log; head -3 $O/shape_profile_precise.log
# the driver's default command, CPU leg included (thread sweep + whole step)
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print('default', d['value'],d['ms_per_step'],d['cpu_baseline'])"
