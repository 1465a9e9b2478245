#!/bin/bash
# round 6: the clock-dependent part of the checkpoint once more (boxes of the pool differ by 5 %): default bench without the CPU leg,
# rocprofv3 kernel stats (one-stream), per-shape profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6z2
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python bench.py --cpu-baseline none > $O/bench_default_no_cpu.json 2> $O/bench_no_cpu.err
python -c "import json;d=json.loads(open('$O/bench_default_no_cpu.json').read().strip().splitlines()[-1]);print('default(no cpu leg)', d['value'],d['ms_per_step'],d['parity']['eps_max_abs_err'], d['roofline']['frac'], d['roofline']['clocks'])"
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r6z_prof -- $BENCH > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
find /tmp/r6z_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
timeout 300 python tools/shape_profile.py precise 2>&1 | grep -v amdgpu.ids > $O/shape_profile_precise.log; head -3 $O/shape_profile_precise.log
timeout 400 python bench.py --cpu-baseline none --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python -c "import json;d=json.loads(open('$O/bench_driver_args.json').read().strip().splitlines()[-1]);print('driver args', d['value'],d['ms_per_step'], d['roofline']['clocks']['sclk_mhz_median'])"
