#!/bin/bash
# round-3 final measurement pass, part 1: rocprofv3 kernel stats, PMC passes (traffic + MFMA busy), per-shape profile, side benches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h
mkdir -p $O
export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r3h_prof -- $BENCH > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
find /tmp/r3h_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
BENCH1="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes"
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- $BENCH1 > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1)
done
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 3 precise "bench.py --steps 1 --warmup 1 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes" > $O/pmc.log 2>&1
python tools/pmc_traffic.py --mfma /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES 3 precise 180.0 >> $O/pmc.log 2>&1
mkdir -p $O/pmc && cp profiles/round3/pmc_* $O/pmc/ 2>/dev/null
tail -14 $O/pmc.log
timeout 300 python tools/shape_profile.py precise 2>&1 | grep -v amdgpu.ids > $O/shape_profile_precise.log
head -3 $O/shape_profile_precise.log
for v in "frames1:--frames 1" "yaml_exact:--yaml-exact" "hoist:--hoist" "ff_chain:--ff-chain"; do
  n=${v%%:*}; a=${v#*:}
  timeout 300 python bench.py $a --cpu-baseline none --no-modes --no-kernel-breakdown 2>$O/side_$n.err | tail -1 > $O/bench_$n.json
  python -c "import json,sys; d=json.loads(open('$O/bench_$n.json').read()); print('$n', d['value'], d['ms_per_step'])"
done
for st in vae-decode vae-encode; do
  timeout 300 python bench.py --stage $st --steps 5 --warmup 2 --no-cpu-baseline 2>$O/side_$st.err | tail -1 > $O/bench_$st.json
  python -c "import json,sys; d=json.loads(open('$O/bench_$st.json').read()); print('$st', d['value'], d['ms_per_step'])"
done
