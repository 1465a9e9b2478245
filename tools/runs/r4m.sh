#!/bin/bash
# round 4, call 13: where the view loop-back's extra time goes (kernel stats of the unsharded and of the loop-back evaluations, apart)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m
mkdir -p $O
export TMPDIR=/tmp
for w in unsharded loopback; do
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -- python $GRAFT_REPO_ROOT/tools/exp/view_loopback_time.py 3 $w > $GRAFT_REPO_ROOT/$O/run_$w.log 2>&1)
grep "ms /" $O/run_$w.log
done
python - <<'PY' | tee $O/view_loopback_kernel_stats.txt
import csv, glob, re
def load(w):
    f = glob.glob(f'/tmp/prof_{w}/**/*kernel_stats.csv', recursive=True)[0]
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs']) / 5e6) for r in csv.DictReader(open(f))}
a, b = load('unsharded'), load('loopback')
print(f"kernel time per evaluation: unsharded {sum(v[1] for v in a.values()):.2f} ms, loop-back {sum(v[1] for v in b.values()):.2f} ms")
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca // 5, ta, cb // 5, tb))
for d, k, ca, ta, cb, tb in sorted(rows, key=lambda r: -abs(r[0]))[:30]:
    print(f"{k[:100]:100s} {ca:5d} {ta:8.3f} ms -> {cb:5d} {tb:8.3f} ms  ({d:+.3f})")
PY
