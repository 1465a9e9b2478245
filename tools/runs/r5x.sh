#!/bin/bash
# round 5: clocks and package power while the bench runs (is the step paced by the sustained clocks?), against short GEMM bursts
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5x
mkdir -p $O
export TMPDIR=/tmp
rocm-smi --showclocks --showpower --showmaxpower 2>&1 | grep -v "^$" > $O/idle.txt
sample() {   # $1 = tag, runs until the file $O/stop exists
  while [ ! -f $O/stop ]; do
    echo "T $(date +%s.%N)" >> $O/smi_$1.log
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" >> $O/smi_$1.log
    sleep 0.15
  done
}
rm -f $O/stop; sample bench &
SP=$!
timeout 300 python bench.py --steps 80 --warmup 5 --cpu-baseline none --no-modes --no-parity --no-kernel-breakdown > $O/bench.json 2> $O/bench.err
touch $O/stop; wait $SP
rm -f $O/stop; sample fast &
SP=$!
timeout 300 python bench.py --precision fast --steps 80 --warmup 5 --cpu-baseline none --no-modes --no-parity --no-kernel-breakdown > $O/bench_fast.json 2> $O/bench_fast.err
touch $O/stop; wait $SP
rm -f $O/stop; sample ksweep &
SP=$!
timeout 200 python tools/exp/stagger_ksweep.py > $O/ksweep.log 2>&1
touch $O/stop; wait $SP
rm -f $O/stop
python - <<'PY'
import re,glob
for f in sorted(glob.glob('gpurun_out/r5x/smi_*.log')):
    s=open(f).read()
    sclk=[int(x) for x in re.findall(r'sclk.*?\((\d+)Mhz\)', s)]
    pw=[float(x) for x in re.findall(r'Power.*?:\s*([\d.]+)', s)]
    if sclk: print(f, 'samples', len(sclk), 'sclk mean', sum(sclk)/len(sclk), 'min', min(sclk), 'max', max(sclk))
    if pw: print(f, 'power mean', sum(pw)/len(pw), 'max', max(pw))
PY
head -30 $O/smi_bench.log | tail -12
