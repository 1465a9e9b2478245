#!/bin/bash
# round 4, call 1: whole -m gpu suite (new pins for configs 2 / 5, full-width sharded layouts, RCCL view transport), attention A/B
# (two workgroups per CU, deferred running max, incremental tile addresses), default bench + whole-step A/Bs, rocprofv3 stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a
mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/test_measurements.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=1500 -x 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/gpu_tests.log
cp gpurun_out/test_measurements.log $O/ 2>/dev/null
timeout 300 python tools/exp/attn_ab.py 2 > $O/attn_ab.log 2>&1
tail -50 $O/attn_ab.log
B="--steps 20 --warmup 3 --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
for rd in 1 2; do
for opt in "" "--set-option ATTN_VARIANT=1" "--set-option ATTN_DEFER_MAX=0" "--set-option ATTN_DMA=2"; do
  tag=$(echo "$opt" | tr -c 'A-Za-z0-9=' '_')
  timeout 400 python bench.py $B $opt > $O/bench_${rd}_$tag.json 2> $O/bench_${rd}_$tag.err
  python -c "import json;d=json.loads(open('$O/bench_${rd}_$tag.json').read().strip().splitlines()[-1]);print('round $rd [$opt]', d['ms_per_step'])"
done
done
timeout 400 python bench.py --cpu-baseline none > $O/bench_default_no_cpu.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench_default_no_cpu.json').read().strip().splitlines()[-1]);print('default', d['value'],d['ms_per_step'],d['parity']['eps_max_abs_err'],d['modes']['fast']['ms_per_step'], d['roofline']['frac'])"
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r4a_prof -- $BENCH > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
find /tmp/r4a_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
