#!/bin/bash
# round 5: which torch (non-library) kernels run PER STEP?  kernel stats of a 1-step and a 3-step run, differenced
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5v
mkdir -p $O
export TMPDIR=/tmp
for n in 1 3; do
  B="python $GRAFT_REPO_ROOT/bench.py --steps $n --warmup 0 --one-stream --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/v$n -- $B > $GRAFT_REPO_ROOT/$O/prof$n.log 2>&1)
  find /tmp/v$n -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/stats$n.csv
done
python - <<'PY'
import csv
a={r['Name']:(int(r['Calls']),int(r['TotalDurationNs'])) for r in csv.DictReader(open('gpurun_out/r5v/stats1.csv'))}
b={r['Name']:(int(r['Calls']),int(r['TotalDurationNs'])) for r in csv.DictReader(open('gpurun_out/r5v/stats3.csv'))}
rows=[]
for k,(c3,t3) in b.items():
    c1,t1=a.get(k,(0,0))
    rows.append(((t3-t1)/2e3,(c3-c1)/2,c1,t1/1e3,k))
rows.sort(reverse=True)
tot=sum(r[0] for r in rows)
print(f"per-step kernel time {tot/1e3:.2f} ms")
print("us/step  calls/step | calls in the 1-step run (incl. setup), us | name")
for us,c,c1,t1,k in rows:
    if 'pnc_gemm' in k or '_GLOBAL__N' in k or 'anonymous' in k: continue
    print(f"{us:9.1f} {c:7.1f} | {c1:5d} {t1:10.1f} | {k[:150]}")
PY
