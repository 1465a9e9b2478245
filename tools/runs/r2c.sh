#!/bin/bash
# third GPU pass: the two tests that met a stale library, bench (both policies), clean rocprofv3 stats, PMC traffic passes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout=600 -k "fused or pipeline or frame_shard or lo_planes or layout" 2>&1 | tail -15 > gpurun_out/r2c_pytest.log
tail -4 gpurun_out/r2c_pytest.log
timeout 400 python bench.py --steps 10 --warmup 2 --cpu-baseline none > gpurun_out/r2c_bench.json 2> gpurun_out/r2c_bench.err
python -c "import json;d=json.loads(open('gpurun_out/r2c_bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['parity'],d['modes']['fast']['ms_per_step'])"
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2c_prof -- $BENCH > $GRAFT_REPO_ROOT/gpurun_out/r2c_prof.log 2>&1)
find gpurun_out/r2c_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2c_kernel_stats.csv
rm -rf gpurun_out/r2c_prof
BENCH1="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- $BENCH1 > $GRAFT_REPO_ROOT/gpurun_out/r2c_pmc_$c.log 2>&1)
done
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 3 precise "bench.py --steps 1 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes" > gpurun_out/r2c_pmc.log 2>&1
mkdir -p gpurun_out/r2c_pmc && cp profiles/round2/pmc_* gpurun_out/r2c_pmc/ 2>/dev/null
tail -15 gpurun_out/r2c_pmc.log
