#!/bin/bash
# first GPU pass of round 2: parity tests, bench (both operand policies), kernel micro-benchmarks, rocprofv3 kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -40 > gpurun_out/r2a_pytest.log
tail -5 gpurun_out/r2a_pytest.log
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-baseline none > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
tail -c 3000 gpurun_out/r2a_bench.json
timeout 400 python tools/kbench.py gemm > gpurun_out/r2a_kbench_gemm.log 2>&1
timeout 300 python tools/kbench.py conv >> gpurun_out/r2a_kbench_gemm.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown > $GRAFT_REPO_ROOT/gpurun_out/r2a_prof.log 2>&1)
find gpurun_out/r2a_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2a_kernel_stats.csv
find gpurun_out/r2a_prof -type f ! -name "*stats*" -delete
head -25 gpurun_out/r2a_kernel_stats.csv
