#!/bin/bash
# round 5: the bench's own clock sampler (sysfs) on the GPU box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5w
mkdir -p $O
ls /sys/class/drm/ > $O/sysfs.txt; for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo $f; cat $f; done >> $O/sysfs.txt 2>&1
ls /sys/class/drm/card*/device/hwmon/hwmon*/ >> $O/sysfs.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 3 --cpu-baseline none --no-modes --no-kernel-breakdown > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('clocks'))"
tail -3 $O/bench.err
