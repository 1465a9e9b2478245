#!/bin/bash
# round 5, call 1: (1) GEMM schedule probe (shipped loop vs phased / staggered / unit-ring), (2) the vendor GEMM's kernel names for the
# main-loop-bound shapes, (3) the two new reference pins (second weight draw, heavy-tail weight set) on the GPU, (4) same-box baseline bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a
mkdir -p $O
export TMPDIR=/tmp
timeout 240 tools/exp/gemm_phase_probe 7 > $O/gemm_phase_probe.log 2>&1; echo "probe rc $?" >> $O/gemm_phase_probe.log
tail -12 $O/gemm_phase_probe.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/vendor -o vendor --output-format csv -- python $GRAFT_REPO_ROOT/tools/exp/vendor_kernel_names.py > $GRAFT_REPO_ROOT/$O/vendor_names.log 2>&1 )
grep SHAPE $O/vendor_names.log
find $O/vendor -name "*kernel_stats.csv" | head -1 | xargs -r cat | cut -c1-400 | head -12 > $O/vendor_kernel_stats.txt
cat $O/vendor_kernel_stats.txt | cut -c1-250
find $O/vendor -name "*kernel_trace.csv" -delete
rm -f gpurun_out/test_measurements.log
timeout 900 python -m pytest -q --timeout=800 tests/test_model_gpu.py -k "full_size" -s 2>&1 | grep -v amdgpu.ids | grep -E "vs reference|passed|failed|Error|error|assert" | tail -20 | tee $O/pins.log
cp gpurun_out/test_measurements.log $O/test_measurements.log 2>/dev/null
timeout 400 python bench.py --steps 20 --warmup 3 --cpu-baseline none --no-modes > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('bench', round(d['ms_per_step'],2), d['roofline']['frac'], [round(p['eps_max_abs_err']*1e4,2) for p in d['parity']['pins']])" | tee $O/bench_summary.txt
