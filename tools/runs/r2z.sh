#!/bin/bash
# measurement pass for the final build (no test suite: r2v ran it on the same code; only comments changed since): default bench line,
# rocprofv3 stats, PMC passes (traffic + MFMA busy)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=300 -k "conv3x3" 2>&1 | tail -2
timeout 400 python bench.py --steps 10 --warmup 2 --cpu-baseline none > gpurun_out/r2z_bench.json 2> gpurun_out/r2z_bench.err
python -c "import json;d=json.loads(open('gpurun_out/r2z_bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['parity']['eps_max_abs_err'],d['modes']['fast']['ms_per_step'], d['roofline']['dominant_kernel']['avg_launch_us'])"
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2z_prof -- $BENCH > $GRAFT_REPO_ROOT/gpurun_out/r2z_prof.log 2>&1)
find gpurun_out/r2z_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2z_kernel_stats.csv
rm -rf gpurun_out/r2z_prof
BENCH1="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes"
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- $BENCH1 > $GRAFT_REPO_ROOT/gpurun_out/r2z_pmc_$c.log 2>&1)
done
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 3 precise "bench.py --steps 1 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes" > gpurun_out/r2z_pmc.log 2>&1
python tools/pmc_traffic.py --mfma /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES 3 precise 190.0 >> gpurun_out/r2z_pmc.log 2>&1
mkdir -p gpurun_out/r2z_pmc && cp profiles/round2/pmc_* gpurun_out/r2z_pmc/ 2>/dev/null
tail -12 gpurun_out/r2z_pmc.log | head -12
